// Wavefront-level primitives (gfx950, wave64) shared by the wave-cooperative kernels.
// All of them must be called with EXEC full (wave-uniform control flow).
#pragma once
#include <hip/hip_runtime.h>

namespace wavep {
typedef unsigned long long u64;

// LDS ordering inside one wave: DS operations execute in order, this only stops the compiler from moving or
// caching LDS accesses across a phase boundary.
__device__ inline void wsync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ inline int rl(int v, int l) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(l)); }
// Marks a value the algorithm keeps wave-uniform as uniform for the compiler as well.  Without it one value that passed
// through a VGPR-only operation (ds_bpermute, a per-lane select) makes every branch that depends on it a divergent one:
// EXEC-mask bookkeeping around each `if`, uniform state held in VGPRs, and s_waitcnt 0 in front of every load.
__device__ inline int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ inline unsigned uni(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ inline bool uni(bool v) { return __builtin_amdgcn_readfirstlane((int)v) != 0; }
__device__ inline u64 lt_mask(int lane) { return (1ull << lane) - 1ull; }
__device__ inline u64 gt_mask(int lane) { return (~0ull << lane) << 1; }
__device__ inline int hibit(u64 m) { return 63 - __builtin_clzll(m); }

struct OpMin { __device__ inline int operator()(int a, int b) const { return min(a, b); } };
struct OpMax { __device__ inline int operator()(int a, int b) const { return max(a, b); } };
struct OpSum { __device__ inline int operator()(int a, int b) const { return a + b; } };
// full-wave reduction to a uniform value: xor-1, xor-2 (quad_perm), row_half_mirror, row_mirror, then the four rows
template <class Op> __device__ inline int wred(int v, Op op) {
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return op(op(a, b), op(c, d));
}
__device__ inline int wmin(int v) { return wred(v, OpMin()); }
__device__ inline int wmax(int v) { return wred(v, OpMax()); }
__device__ inline int wsum(int v) { return wred(v, OpSum()); }
__device__ inline int popc(u64 m) { return __builtin_popcountll(m); }


}  // namespace wavep
