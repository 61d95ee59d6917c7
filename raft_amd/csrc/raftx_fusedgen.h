// raftx_fusedgen.h -- the generating form of the persistent fused fixed point (included by raftx_hip.hip after
// raftx_kernels.h and raftx_geom.h).
//
// In a streamed sweep (raftx_sweep_prepare / _launch / _wait) the strip tables of batch i+1 used to be written by a kernel
// of their own (k_geom_design + k_geom_addup) that can only get onto the chip when the fused kernel of batch i drains:
// ~0.3 ms per step in which the chip runs latency-bound small kernels at a few waves per CU and the next fused kernel
// waits ("between two fused kernels").  Here the workgroup that has CLAIMED a pair builds that design's
// tables itself, right before it solves the pair: the same device function (geom_design_block: strip records from the
// member pass's poses, run detection, device records, Morison added mass, the design's share of the add-up), its loads
// and stores hidden behind the seven other waves of the CU, and the next fused kernel needs nothing that is not already
// there when the one before it ends.  The tables are written to HBM as before (the fixed point reads them through the
// scalar cache, raftx_fetch_* read them afterwards); the ABI copy of the strip records is not (sweeps never fetch it).
//
// Valid for launches in which every design is claimed exactly once (one sea state per design, no pair list) and without
// MacCamy-Fuchs rows (their table is a kernel of its own behind the generation); the host takes k_geom_design otherwise.
#pragma once

struct PersistGenArgs {
    PersistArgs P;
    GeomArgs G;
};

// What the workgroup has just stored is read back by the SAME workgroup only: through the vector L1 of its CU (M0, C0;
// coherent for the waves of a CU once the stores have left the wave) and through the scalar cache (strip records, flags:
// constant address space), which is not coherent with vector stores and is dropped -- lines of these addresses may
// survive from a neighbour's read of a line this design shares (flags: 4 B per strip).  Workgroup scope on purpose: an
// agent-scope release / acquire pair writes back and invalidates the whole L2 of the XCD on gfx950 -- once per design
// that cost the kernel 0.5 ms per 10 000 (profiles/r06_experiments/fused_generation_ab.txt).
__device__ __forceinline__ int opaque_uniform(int x) {       // opaque() for a wave-uniform value: it stays in a scalar register
    asm volatile("" : "+s"(x));
    return x;
}
__device__ __forceinline__ void fusedgen_publish() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every store of this wave acknowledged by the L2 (the scalar loads
                                                          // that follow go there, not through this CU's vector L1)
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

#define RAFTX_KPG_DEFINE(NAME, NB, FLAGS, MAXT, MINB)                                                                   \
    extern "C" __global__ void __launch_bounds__(MAXT, MINB) NAME(PersistGenArgs PG) {                                  \
        extern __shared__ __attribute__((aligned(16))) double smem[];                                                   \
        static_assert(MAXT == GD_T, "the design's tables are built by the pair's own workgroup");                       \
        if (threadIdx.x == 0) {                                                                                         \
            LDS_AS unsigned *st_ = (LDS_AS unsigned *)smem;                                                             \
            const size_t qp_ = (size_t)__builtin_amdgcn_queue_ptr(), ka_ = (size_t)__builtin_amdgcn_kernarg_segment_ptr(); \
            st_[1] = blockIdx.x + 0u * (blockIdx.y + blockIdx.z + threadIdx.y + threadIdx.z);                           \
            st_[2] = (unsigned)qp_;                                                                                     \
            st_[3] = (unsigned)(qp_ >> 32);                                                                             \
            st_[4] = (unsigned)ka_;                                                                                     \
            st_[5] = (unsigned)(ka_ >> 32);                                                                             \
        }                                                                                                               \
        const int nl = PG.P.T.nDesign;                                       /* one sea state per design, no pair list */ \
        int idx = claim_pair(PG.P.ctr, nl, (LDS_AS int *)smem);                                                         \
        if (idx < 0) {                                                                                                  \
            kp_leave(PG.P);                                                                                             \
            return;                                                                                                     \
        }                                                                                                               \
        geom_design_block<false, true>(PG.G, idx, smem + KP_STASH, reinterpret_cast<int *>(smem) + 6);                  \
        fusedgen_publish();                                                                                             \
        idx = opaque_uniform(idx);                             /* every table address of the pair is formed behind this point */ \
        solve_pair<NB, FLAGS, MAXT, true>(PG.P.T, PG.P.A, idx, PG.P.xl_base + blockIdx.x);                              \
        RAFTX_KP_REENTER(NAME);                                                                                         \
    }
RAFTX_KPG_DEFINE(raftx_kpg_f0, 2, 0, 128, RAFTX_KP_MINB)
