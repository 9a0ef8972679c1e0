"""Ulysses sequence parallelism for ONE denoise chain over P GPUs.

What the reference does (wan23/distributed/ulysses.py:9-47, sequence_parallel.py:65-176, util.py:21-31): every rank keeps
L/P tokens for all per-token work; around the self-attention it runs FOUR all-to-alls per block (q, k, v: scatter heads,
gather sequence; output: the reverse), each a list all_to_all + cat, on token-major [B, L/P, N, D] tensors.

Here, for RCCL over point-to-point xGMI links (one message per peer per collective, per-link bound):
  * TWO collectives per block instead of four: q, k and the K-major V^T slab of every destination rank travel in ONE
    `all_to_all_single` (3 * Lp * C / P bf16 elements per peer), the attention output in a second one;
  * the operands arrive in exactly the layouts the attention kernel consumes (yume_attn_fwd: q, k token-major
    [P*Lp, C/P], V^T K-major [C/P, P*Lp]) — V is never transposed again after the QKV GEMM epilogue wrote it K-major;
  * the per-rank chunk length Lp is ceil(L / P) rounded up to 8 (16-byte rows of the K-major image); pad tokens sit at
    the global end and are masked as keys by passing the true L as the key count.
Heads must divide by P (24 = 5B, 40 = 14B: P in {1, 2, 4, 8}). Everything else in the block (adaLN, GEMMs, RMSNorm +
RoPE, cross-attention over the full context, FFN) is row-local and runs on the rank's Lp tokens unchanged.

torch.distributed is plumbing: backend "nccl" (= RCCL) moves device buffers directly; with the "gloo" backend (CPU tests,
and two ranks sharing one GPU in the parity test) the same buffers are staged through host memory.
"""
import torch
import torch.distributed as dist


def _round_up(a, b):
    return (a + b - 1) // b * b


class SequenceParallel:
    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("sequence parallelism needs an initialised torch.distributed process group")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.host_staged = dist.get_backend(group) == "gloo"

    # ------------------------------------------------------------------ partition
    def chunk(self, L):
        """-> (Lp, lo, hi): this rank owns global tokens [lo, hi) and computes on Lp rows (rows hi-lo.. are padding)."""
        Lp = _round_up(-(-L // self.world), 8)
        lo = min(self.rank * Lp, L)
        hi = min(lo + Lp, L)
        return Lp, lo, hi

    def check_heads(self, H):
        if H % self.world:
            raise RuntimeError(f"sequence parallelism over {self.world} ranks needs the head count ({H}) to divide evenly")

    # ------------------------------------------------------------------ collectives
    def _all_to_all(self, send):
        """send [P, n] (row j goes to rank j) -> recv [P, n] (row i came from rank i)."""
        recv = torch.empty_like(send)
        if self.world == 1:
            recv.copy_(send)
        elif self.host_staged:
            s = send.cpu().contiguous()
            r = torch.empty_like(s)
            dist.all_to_all_single(r.view(torch.uint8), s.view(torch.uint8), group=self.group)   # gloo moves bytes
            recv.copy_(r)
        else:
            dist.all_to_all_single(recv, send, group=self.group)
        return recv

    def exchange_qkv(self, qk, vt, C):
        """qk bf16 [Lp, 2C] (q | k, this rank's tokens, all heads); vt bf16 [C, Lp] (K-major V^T).
        -> q, k bf16 [P*Lp, C/P] and vt bf16 [C/P, P*Lp]: ALL tokens (global order), this rank's heads."""
        P = self.world
        Lp = qk.shape[0]
        Cp = C // P
        n = Lp * Cp
        send = torch.empty((P, 3 * n), dtype=qk.dtype, device=qk.device)
        sv = send.view(P, 3, n)
        # destination-major packing: q, k as [Lp, Cp] blocks, V^T as the destination's [Cp, Lp] row slab
        sv[:, 0].view(P, Lp, Cp).copy_(qk[:, :C].view(Lp, P, Cp).transpose(0, 1))
        sv[:, 1].view(P, Lp, Cp).copy_(qk[:, C:].view(Lp, P, Cp).transpose(0, 1))
        sv[:, 2].view(P, Cp, Lp).copy_(vt.view(P, Cp, vt.shape[1])[:, :, :Lp])
        recv = self._all_to_all(send).view(P, 3, n)
        q = recv[:, 0].reshape(P * Lp, Cp)                              # source-major rows = global token order
        k = recv[:, 1].reshape(P * Lp, Cp)
        v = recv[:, 2].view(P, Cp, Lp).transpose(0, 1).reshape(Cp, P * Lp)   # columns: source rank, then its tokens
        return q, k, v

    def exchange_out(self, o):
        """o bf16 [P*Lp, C/P] (all tokens, this rank's heads) -> bf16 [Lp, C] (this rank's tokens, all heads)."""
        P = self.world
        Lp, Cp = o.shape[0] // P, o.shape[1]
        recv = self._all_to_all(o.view(P, Lp * Cp))                     # row i: my tokens, heads of rank i
        return recv.view(P, Lp, Cp).transpose(0, 1).reshape(Lp, P * Cp)

    def gather_rows(self, y):
        """y [Lp, n] -> [P*Lp, n] in global token order on every rank (the reference's gather_forward, util.py:42-51)."""
        if self.world == 1:
            return y
        out = torch.empty((self.world * y.shape[0], y.shape[1]), dtype=y.dtype, device=y.device)
        if self.host_staged:
            h = torch.empty(out.shape, dtype=y.dtype)
            dist.all_gather_into_tensor(h.view(torch.uint8), y.cpu().contiguous().view(torch.uint8), group=self.group)
            out.copy_(h)
        else:
            dist.all_gather_into_tensor(out, y.contiguous(), group=self.group)
        return out
