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

// Full-wave reduction to a uniform value, six DPP steps and one readlane: xor-1, xor-2 (quad_perm), row_half_mirror and
// row_mirror leave every lane of a row with the row's result; row_bcast:15 folds row 0 into row 1 and row 2 into row 3,
// row_bcast:31 folds rows 0-1 into rows 2-3, lane 63 holds the wave's result.  Written as one asm block because the
// compiler expands each update_dpp + op into mov / nop / mov_dpp / op (26 instructions per reduction instead of 13);
// the s_nop 1 in front of each step is the "VALU write -> DPP read" wait the hazard recogniser cannot add inside asm.
#define WAVEP_RED(NAME, OP)                                                                             \
    __device__ inline int NAME(int v) {                                                                 \
        asm("s_nop 1\n\t" OP "_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"         \
            "s_nop 1\n\t" OP "_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"         \
            "s_nop 1\n\t" OP "_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"             \
            "s_nop 1\n\t" OP "_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"                  \
            "s_nop 1\n\t" OP "_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                \
            "s_nop 1\n\t" OP "_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"                      \
            : "+v"(v));                                                                                 \
        return __builtin_amdgcn_readlane(v, 63);                                                        \
    }
WAVEP_RED(wmin, "v_min_i32")
WAVEP_RED(wmax, "v_max_i32")
WAVEP_RED(wsum, "v_add_u32")
#undef WAVEP_RED
// lane masks straight from one v_cmp (EXEC is full, so the mask needs no "and with EXEC" round trip)
__device__ inline u64 mask_eq(int a, int b) { return __builtin_amdgcn_sicmp(a, b, 32); }
__device__ inline u64 mask_ule(unsigned a, unsigned b) { return __builtin_amdgcn_uicmp(a, b, 37); }
__device__ inline int popc(u64 m) { return __builtin_popcountll(m); }


}  // namespace wavep
