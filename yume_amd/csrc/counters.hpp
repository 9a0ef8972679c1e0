// counters.hpp — in-kernel ticket counters live in CALLER-OWNED device memory ("the library never allocates persistent device
// memory except an explicit yume_workspace_bytes() / ..._init(ptr) pair").
//
// yume_counter_workspace_init(ptr, bytes, stream) registers, for the calling thread's current device, a buffer of
// yume_counter_workspace_bytes() bytes that the call zeroes (on `stream`) and that must stay valid until it is unregistered (ptr = NULL) or
// replaced. The buffer is SETS sets of 64 bytes. A launch that hands out work by ticket (conv_w4.hpp: the tails of long convolutions;
// attn_fwd8.hip: every item of the persistent attention kernel) takes the next set round-robin; the invariant is that a set holds zeros
// whenever no launch is using it — the LAST workgroup of a launch to touch its set writes the zeros back — so there is no memset in front
// of a launch (capturable into a hipGraph), and a set is only shared by two launches if SETS (256) ticketed launches of one device are in
// flight at once, on any number of streams. Without a registered buffer next_set() returns nullptr and the callers keep their static
// (ticket-free) schedules.
#pragma once

namespace yume_counters {
constexpr int SET_INTS = 16;        // one set = 64 bytes: up to 16 counters
constexpr int SETS = 256;
int* next_set();                    // the next set of the current device's registered workspace, or nullptr
}  // namespace yume_counters
