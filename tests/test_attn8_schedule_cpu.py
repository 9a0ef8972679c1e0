"""The control protocol of the persistent attention kernel (yume_amd/csrc/attn_fwd8.hip), transcribed: the item loop of ONE workgroup — lazy
tickets, the K / V^T stream that runs 4 / 3 tiles ahead of the tile being computed and wraps into the next item, the slot of a tile
(global tile counter & 3), the tile kinds (steady / first-behind-a-boundary / masked / boundary / final) — walked over item sequences,
with the invariants the kernel relies on asserted at every step:
  * the tile a step computes finds its V^T(t) and K(t+1), K(t+2) in the slots it reads, fetched at least one step earlier;
  * a slot is never overwritten before its last reader;
  * the next item is known (its ticket read behind a barrier) before the stream wraps into it;
  * every ticket is drawn once, the exhausted-queue ticket exactly once per workgroup.
A model of the control flow only (no arithmetic): it fails when an edit of the loop breaks the protocol, long before a GPU run would."""
import itertools

import pytest


class Queue:
    def __init__(self, items):
        self.items, self.next, self.empty_draws = list(items), 0, 0

    def draw(self):
        if self.next < len(self.items):
            self.next += 1
            return self.items[self.next - 1]
        self.empty_draws += 1
        return None


def run_workgroup(q, log):
    """items: (name, n_tiles). Mirrors the structure of attn_fwd_kernel_v8's `for (;;)` loop."""
    cur = q.draw()
    if cur is None:
        return
    nxt, have_nxt, drew, mail = None, False, False, None
    kslot, vslot = {}, {}                  # slot -> (item, tile, step fetched)
    step = 0                               # one "step" = one tile of compute (one barrier)
    cold, first = True, False
    g = t = kleft = vleft = 0
    kstream = vstream = None               # (item, next tile to fetch)
    while True:
        name, n = cur
        assert n >= 5
        if cold:
            cold = False
            g, t = 0, 0
            for i in range(4):
                kslot[i] = (name, i, step - 1)
            for i in range(3):
                vslot[i] = (name, i, step - 1)
            kstream, vstream = [cur, 4], [cur, 3]
            kleft, vleft = n - 4, n - 3
            first = False
        rem = n - t

        def next_ticket():
            nonlocal nxt, have_nxt, drew, mail
            if have_nxt:
                return
            if drew:
                nxt, have_nxt, drew = mail, True, False
            elif rem <= 8:
                mail, drew = q.draw(), True

        def need_nxt():
            nonlocal nxt, have_nxt, drew, mail
            if have_nxt:
                return
            log.append("need_nxt fallback")
            if not drew:
                mail = q.draw()
            nxt, have_nxt, drew = mail, True, False

        def wraps():
            nonlocal kleft, vleft, kstream, vstream
            if kleft == 0:
                need_nxt()
                kstream, kleft = ([nxt, 0], nxt[1]) if nxt else ([None, 0], 1 << 28)
            if vleft == 0:
                need_nxt()
                vstream, vleft = ([nxt, 0], nxt[1]) if nxt else ([None, 0], 1 << 28)

        def tile(kind):
            """one tile step at global counter g computing tile t of cur; issues K(t+4) -> slot g & 3, V^T(t+3) -> slot (g + 3) & 3"""
            nonlocal g, t, rem, kleft, vleft, step
            # what it reads: V^T(t) in slot g & 3; K(t+2) in slot (g + 2) & 3 (cache refill) unless this is the final tile
            it, tt, when = vslot[g & 3]
            assert (it, tt) == (name, t) and when < step, (kind, name, t, vslot)
            if kind != "final":
                want = (name, t + 2) if t + 2 < n else ((nxt[0], t + 2 - n) if nxt else None)
                if want is not None:
                    it, tt, when = kslot[(g + 2) & 3]
                    assert (it, tt) == want and when < step, (kind, name, t, want, kslot)
                # its own fetches: into the slot of K(t) (consumed two steps ago) and of V^T(t-1) (consumed in the previous step)
                if kstream[0] is not None:
                    kslot[g & 3] = (kstream[0][0], kstream[1], step)
                    kstream[1] += 1
                if vstream[0] is not None:
                    vslot[(g + 3) & 3] = (vstream[0][0], vstream[1], step)
                    vstream[1] += 1
            log.append((name, t, kind))
            step += 1
            g += 1
            t += 1
            rem -= 1
            kleft -= 1
            vleft -= 1

        next_ticket()
        wraps()
        if first:
            first = False
            tile("first")
            next_ticket()
            wraps()
        while rem > 2:
            if (g & 3) == 1 and rem >= 10 and kleft >= 4 and vleft >= 4:
                while True:
                    for _ in range(4):
                        tile("steady")
                    if not (rem >= 10 and kleft >= 4 and vleft >= 4):
                        break
            else:
                tile("steady")
            next_ticket()
            wraps()
        need_nxt()
        tile("masked")
        wraps()
        if nxt:
            tile("boundary")
            t -= 1              # (the kernel does not advance t behind the boundary tile; it is reset below)
        else:
            tile("final")
        # bubble 2: cur's O^T is stored here
        log.append((name, "stored"))
        if not nxt:
            break
        cur, have_nxt, drew, nxt = nxt, False, False, None
        t, first = 0, True


@pytest.mark.parametrize("lengths", [[148] * 5, [8] * 9, [8, 148, 9, 37, 8, 74, 74], [11, 10, 9, 8], [24] * 3, [5, 6, 7, 5, 12, 5]])
def test_one_workgroup_walks_its_items(lengths):
    items = [(f"i{k}", n) for k, n in enumerate(lengths)]
    q, log = Queue(items), []
    run_workgroup(q, log)
    assert q.empty_draws == 1                                          # the exhausted-queue ticket: exactly once
    assert "need_nxt fallback" not in log                              # the lazy ticket was always there in time (items >= 5 tiles)
    for name, n in items:
        tiles = [e for e in log if e[0] == name and e[1] != "stored"]
        assert [e[1] for e in tiles] == list(range(n))                 # every tile once, in order
        kinds = [e[2] for e in tiles]
        assert kinds[-2] == "masked" and kinds[-1] in ("boundary", "final")
        assert all(k in ("steady", "first") for k in kinds[:-2]) and kinds.count("first") <= 1
    assert [e[0] for e in log if e[1:] == ("stored",)] == [n for n, _ in items]
    assert [e[2] for e in log if e[1] != "stored"][-1] == "final"


def test_several_workgroups_share_a_queue_without_losing_or_repeating_items():
    """tickets are drawn in whatever order the workgroups reach their draws; interleave four of them step by step"""
    import threading
    items = [(f"i{k}", n) for k, n in enumerate(itertools.islice(itertools.cycle([148, 8, 37, 74, 9]), 23))]
    q = Queue(items)
    lock, logs = threading.Lock(), [[] for _ in range(4)]
    real_draw = q.draw
    q.draw = lambda: (lock.acquire(), real_draw(), lock.release())[1]
    ts = [threading.Thread(target=run_workgroup, args=(q, logs[i])) for i in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    stored = sorted(e[0] for lg in logs for e in lg if e[1:] == ("stored",))
    assert stored == sorted(n for n, _ in items) and q.empty_draws == 4
