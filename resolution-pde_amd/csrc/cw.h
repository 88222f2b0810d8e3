// Counted waits.  gfx950 has ONE in-order counter (vmcnt) for loads, stores and LDS-DMA, and `s_waitcnt vmcnt(N)` returns
// once all but the wave's N youngest vector-memory instructions are done.  The kernels that keep this counter by hand
//  rely, at every such wait, on "at least N vector-memory instructions have been ISSUED behind
// the data I am about to touch" -- an invariant of the COMPILED code (how many stores the compiler emits, whether it
// spills, whether a predicated store sits behind a branch), not of the source.  So both ends are named in the
// instruction stream and tests/test_isa_counted_waits_cpu.py checks the invariant on the gfx950 assembly, over every
// path: between `cw_mark T` and the next `cw_wait T N` at least N certain vector-memory instructions.
//
//   cw_mark<T>()      right behind the LAST vector-memory instruction whose completion a later wait on tag T needs
//                     (everything older completes with it: the counter is in order)
//   cw_wait<T, N>()   s_waitcnt vmcnt(N) for the data marked T
// A tag must not be marked again before it has been waited for (tags = ring slots / buffer parities).
#pragma once

namespace rpde {

template <int TAG>
__device__ __forceinline__ void cw_mark() {
  asm volatile("; cw_mark %0" ::"n"(TAG) : "memory");
}
template <int TAG, int N>
__device__ __forceinline__ void cw_wait() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%1) ; cw_wait %0 %1" ::"n"(TAG), "n"(N) : "memory");
}

// the same with the tag as a value (0..2) that is a constant after unrolling: the branches fold away
template <int N>
__device__ __forceinline__ void cw_wait_t(int tag) {
  if (tag == 0) cw_wait<0, N>(); else if (tag == 1) cw_wait<1, N>(); else cw_wait<2, N>();
}
__device__ __forceinline__ void cw_mark_t(int tag) {
  if (tag == 0) cw_mark<0>(); else if (tag == 1) cw_mark<1>(); else cw_mark<2>();
}

}  // namespace rpde
