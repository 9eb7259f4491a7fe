// resco_sim.hip -- MI355X (gfx950 / CDNA4) batched traffic-signal microsimulator behind the C ABI of
// include/resco_sim.h.  Written for gfx950 only: wave64, LDS-resident environment state, one workgroup
// per environment instance, every tick of an env-step fused into ONE kernel launch.
//
// Hot path replaced (RESCO, paths relative to its repository):
//   MultiSignal.step                 resco_benchmark/multi_signal.py:164-197
//   Signal.prep_phase / set_phase    resco_benchmark/traffic_signal.py:176-187
//   sumo.simulationStep() x 10       resco_benchmark/multi_signal.py:102-105   (SUMO itself: [SUMO-K])
//   Signal.observe / get_vehicles    resco_benchmark/traffic_signal.py:189-247
//   states.drq_norm / mplight / wave resco_benchmark/states.py:34-127
//   rewards.wait / wait_norm / pressure  resco_benchmark/rewards.py:6-41
//
// Layout
//   HBM  : env-major SoA, field[env][slot]; a workgroup streams its env's slab in once per env-step
//          (coalesced, slot-contiguous), keeps it in LDS for all ticks, and streams it out once.
//   LDS  : per-vehicle nodes / arrays + list heads per 64 m cell + approach registers + per-lane aggregates.
//   L2/IC: read-only scenario tables shared by all environments (< 1 MB).
// There is no dense contraction on this path: no MFMA.  Arithmetic is IEEE fp32 with contraction off so the
// CPU oracle (oracle/resco_oracle.c, test-only) reproduces every value bit-for-bit.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "resco_sim.h"

// ---- what resco_step.h asks of its includer (the GPU flavour: LDS atomics, one thread per call of a phase)
#define RS_DEV __device__ __forceinline__
#define RS_HD __host__ __device__
#define RS_MEM __device__ __forceinline__
#define RS_G(p) (p)
#define RS_CARVE __host__ __device__ __forceinline__
__device__ __forceinline__ void rs_atomic_min(int32_t *p, int32_t v) { atomicMin(p, v); }
__device__ __forceinline__ void rs_atomic_min(uint32_t *p, uint32_t v) { atomicMin(p, v); }
__device__ __forceinline__ void rs_atomic_max(int32_t *p, int32_t v) { atomicMax(p, v); }
__device__ __forceinline__ void rs_atomic_add(int32_t *p, int32_t v) { atomicAdd(p, v); }
__device__ __forceinline__ int32_t rs_atomic_fetch_add(int32_t *p, int32_t v) { return atomicAdd(p, v); }
// the lanes of a wave that are here together take consecutive tickets from ONE atomic (the counter is the same for all of them)
__device__ __forceinline__ int32_t rs_wave_ticket(int32_t *p) {
    const unsigned long long m = __ballot(1);
    const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    int32_t base = 0;
    if (below == 0) base = atomicAdd(p, (int32_t)__popcll(m));
    return __builtin_amdgcn_readlane(base, __ffsll((long long)m) - 1) + below;
}
// one atomic per wave instead of one per lane (64 lanes adding to ONE LDS address are served one after the other): called
// where the whole wave is converged
__device__ __forceinline__ void rs_wave_add(int32_t *p, int32_t v) {
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(p, v);
}
__device__ __forceinline__ void rs_wave_max(int32_t *p, int32_t v) {
    for (int m = 32; m > 0; m >>= 1) { const int32_t o = __shfl_xor(v, m); v = o > v ? o : v; }
    if ((threadIdx.x & 63) == 0) atomicMax(p, v);
}
__device__ __forceinline__ void rs_atomic_or(uint32_t *p, uint32_t v) { atomicOr(p, v); }
__device__ __forceinline__ uint32_t rs_atomic_fetch_or(uint32_t *p, uint32_t v) { return atomicOr(p, v); }
__device__ __forceinline__ int rs_atomic_inc(int32_t *p) { return atomicAdd(p, 1); }
__device__ __forceinline__ void rs_atomic_and(uint32_t *p, uint32_t v) { atomicAnd(p, v); }
__device__ __forceinline__ uint32_t rs_atomic_cas(uint32_t *p, uint32_t cmp, uint32_t v) { return atomicCAS(p, cmp, v); }
__device__ __forceinline__ int rs_ffsll(unsigned long long x) { return __ffsll(x); }
__device__ __forceinline__ int rs_clzll(unsigned long long x) { return __clzll((long long)x); }
__device__ __forceinline__ int rs_ffs(uint32_t x) { return __ffs((int)x); }
__device__ __forceinline__ int rs_popc(uint32_t x) { return __popc(x); }
__device__ __forceinline__ float rs_int_as_float(int x) { return __int_as_float(x); }
__device__ __forceinline__ int rs_float_as_int(float x) { return __float_as_int(x); }
__device__ __forceinline__ uint16_t rs_f2h(float x) { return __half_as_ushort(__float2half(x)); }

// a / b by the hardware's Newton sequence without the range scaling (resco_step.h: RS_DIV).  Identical to `/` -- the same v_rcp_f32 and the same
// seven operations -- for finite non-zero b, a = 0 or 2^-100 < |a|, |a / b| and |1 / b| normal: every division of the step kernel
#ifndef RS_IEEE_DIV
__device__ __forceinline__ float rs_div_unscaled(float a, float b) {
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float r1 = __builtin_fmaf(__builtin_fmaf(-b, r0, 1.0f), r0, r0);
    const float q0 = a * r1;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-b, q0, a), r1, q0);
    return __builtin_fmaf(__builtin_fmaf(-b, q1, a), r1, q1);
}
#define RS_DIV(a, b) rs_div_unscaled((a), (b))
#endif
// all four dwords of a Node are "used": the compiler reads them with one ds_read_b128 instead of narrowing the read to the fields a loop
// body happens to need (resco_step.h: node_load)
#define RS_OPAQUE_S(x) asm volatile("" : "+s"(x));
#ifndef RS_NARROW_NODE
#define RS_KEEP4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#endif
extern __shared__ __attribute__((aligned(16))) char rs_smem[];      // THE working memory of a workgroup (dynamic LDS)
#define RS_SMEM rs_smem

#ifdef RS_STUDY_SECTIONS       // study build: where inside the long path of the plan a wave's time goes (tools/phase_profile.py --sections)
__device__ unsigned long long *g_sec_prof;
#define RS_SEC_BEGIN unsigned long long sec_t = wall_clock64();
#define RS_SEC(id) { const unsigned long long sec_1 = wall_clock64(); if (g_sec_prof && (threadIdx.x & 63) == 0 && (blockIdx.x & 15) == 0) atomicAdd(&g_sec_prof[id], (sec_1 - sec_t) + (1ull << 40)); sec_t = wall_clock64(); }
#endif
#ifdef RS_DEVICE_ASSERT        // the CHECKING build (resco_amd/build.py: libresco_sim_check.so): the classification invariants of the step kernel -- a vehicle
// without FL_H never needs the walk over the links, one without FL_MH never leaves its lane -- are counted per environment in
// rs_stats()[11] instead of compiled out; tests/test_gpu_parity.py::test_device_invariant_counter holds the count at zero.  They rest
// on classify() and the plan evaluating the same floating-point expressions to the same bits at different inline sites
// (-ffp-contract=off): the one place where a compiler upgrade could silently skip a stop line.
#define RS_ASSERT(c) if (!(c)) rs_atomic_add(&L.sc[SC_STATS + ST_INVARIANT], 1);     // (`L`: the working memory, in scope at every site)
#endif
#include "resco_step.h"
#include "resco_policy.h"

// ------------------------------------------------------------------------------------------------ kernels
// The tables / state / output descriptors live in ONE constant block in device memory (StepArgs): passed by value they
// would pin ~70 SGPRs for the whole kernel (beyond ~100 the compiler spills SGPRs into VGPR lanes around every use);
// behind a const __restrict__ pointer every field is a re-loadable scalar load.
struct StepArgs { KTab T; State G; Out O; Lds L; };
// the block is read through the CONSTANT address space: scalar loads, and the compiler takes pointers loaded from it for
// global ones (global_load instead of flat_load, which would also tie up the LDS wait counter)
typedef const __attribute__((address_space(4))) StepArgs *StepArgsPtr;

// ---- the long code paths of a tick as FUNCTIONS: measured in round 6 and NOT adopted (build with -DRS_CALL_LONG to get it).  Inlined
// into the tick loop, the walk over the links, the lane-change searches and the hand-over keep ~190 scalars alive at once -- table
// bases, layout offsets, parameters -- and the 64-VGPR build (80 SGPRs: eight waves per SIMD) spills ~110 of them into VGPR lanes, a
// v_readlane in front of every use.  A called function has a register allocation of its own (41 / 59 / 41 VGPRs, 70-78 SGPRs, no
// SGPR spills inside) and loads what it needs from the constant block when it is entered; its arguments arrive in VGPRs (the calling
// convention knows nothing about uniform values) and are made scalars again with v_readfirstlane.  Bit-exact, but 1-2 % SLOWER
// (profiles/r06_ab_call.txt: ingolstadt21 x 4096 3.105 against 3.137 M env-steps/s, cologne1 x 1024 5.70 / 5.77, cologne8 x 2048 7.37 /
// 7.51): the kernel itself still spills 102 scalars (the short paths, C and the observe phases hold as many), the calls add 48 bytes
// of scratch per lane for the callee-saved registers, and a v_readlane is cheap next to what a spill costs elsewhere.
__device__ __forceinline__ int rs_uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ uint32_t rs_uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ float rs_uni(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }
__device__ __forceinline__ StepArgsPtr rs_uni(StepArgsPtr p) {
    const unsigned long long v = (unsigned long long)p;
    return (StepArgsPtr)(((unsigned long long)rs_uni((uint32_t)(v >> 32)) << 32) | rs_uni((uint32_t)v));
}
// f(T, L, G) with the views a kernel of capacity CAP works with
template <int CAP, class F> __device__ __forceinline__ void rs_with_tables(StepArgsPtr Ac, F f) {
    const StepArgs *A = (const StepArgs *)Ac;
    if constexpr (CAP != 0) { const LdsFix<CAP> Lf(A->L); f(A->T, Lf, A->G); }
    else f(A->T, A->L, A->G);
}
// REGS only separates the instantiations by the register budget of the kernel that calls them (64 / 80 / 128 VGPRs)
template <int CAP, int REGS> __device__ __attribute__((noinline)) void rs_fn_plan_long(StepArgsPtr Ac, uint32_t seed, float sigma, int genv, int env, int t, uint32_t tag, int s) {
    Ac = rs_uni(Ac); seed = rs_uni(seed); sigma = rs_uni(sigma); genv = rs_uni(genv); env = rs_uni(env); t = rs_uni(t); tag = rs_uni(tag);
    rs_with_tables<CAP>(Ac, [&](const KTab &T, const auto &L, const State &G) {
        KParams P{};
        P.seed = seed; P.sigma = sigma;
        const int C = CAP ? CAP : T.capacity;
        phase_plan<true>(T, L, Grid{(uint16_t *)L.grid, tag}, G, (size_t)env * C, P, genv, t, s);
    });
}
template <int CAP, int REGS> __device__ __attribute__((noinline)) void rs_fn_lc_decide(StepArgsPtr Ac, int env, int t, uint32_t tag, int s) {
    Ac = rs_uni(Ac); env = rs_uni(env); t = rs_uni(t); tag = rs_uni(tag);
    rs_with_tables<CAP>(Ac, [&](const KTab &T, const auto &L, const State &G) {
        const int C = CAP ? CAP : T.capacity;
        lc_decide_and_flag(T, L, Grid{(uint16_t *)L.grid, tag}, G, (size_t)env * C, t, s);
    });
}
template <int CAP, int REGS> __device__ __attribute__((noinline)) uint32_t rs_fn_move_long(StepArgsPtr Ac, uint32_t out_mask, int env, int t, uint32_t tag, int flags, int s) {
    Ac = rs_uni(Ac); out_mask = rs_uni(out_mask); env = rs_uni(env); t = rs_uni(t); tag = rs_uni(tag); flags = rs_uni(flags);
    int active = 0, halted = 0, top = 0;
    rs_with_tables<CAP>(Ac, [&](const KTab &T, const auto &L, const State &G) {
        KParams P{};
        P.out_mask = out_mask;
        const int C = CAP ? CAP : T.capacity;
        phase_move<true>(T, L, Grid{(uint16_t *)L.grid, tag}, G, P, env, (size_t)env * C, t, (flags & 1) != 0, (flags & 2) != 0, s, active, halted, top);
    });
    return (uint32_t)active | ((uint32_t)halted << 1) | ((uint32_t)top << 2);
}

// grid = n_envs workgroups (one environment each); blockDim.x = 64 * waves (<= 1024), normally one thread per slot.
// PROF: the build with the in-kernel timers (rs_phase_profile); the production kernels carry none of that code
template <bool PROF, int CAP, int REGS> struct DevExec {
    int B;
    int wave;                       // threadIdx.x / 64, wave-uniform: lives in a scalar register
    unsigned long long *prof;       // optional per-phase timers (rs_phase_profile)
    unsigned long long t0;
    StepArgsPtr Ac;
    template <class F> __device__ __forceinline__ void phase(int id, F f) {
        // The thread index is RECOMPUTED per phase from the wave's index (a scalar) and the lane's position in the wave (two VALU
        // instructions): kept in a register across the kernel it costs a VGPR the 64-VGPR build does not have (it lived in scratch
        // and was re-loaded at the top of every phase), and an index the compiler can see through has every address derived from
        // it (the small strided loops of the tick's phases) computed once before the tick loop and kept alive across it --
        // 18 VGPRs spilled to scratch (round 2).
        int w = wave;
        uint32_t ones = ~0u;
        asm volatile("" : "+s"(w), "+s"(ones));     // (opaque: neither the lane index nor anything derived from it is hoisted out of the phase)
        int tid = (w << 6) | (int)__builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u));
        asm volatile("" : "+v"(tid));
        f(tid);
        __syncthreads();
        if (PROF && prof) {         // (t0 stays wave-uniform: every thread takes the time, thread 0 adds it up)
            const unsigned long long t1 = wall_clock64();
            if (tid == 0) atomicAdd(&prof[id], t1 - t0);
            t0 = t1;
        }
    }
    // tid / 64 as a wave-uniform value: what is derived from it (the roles of the waves inside a phase) stays in scalar registers
    __device__ __forceinline__ int wave_of(int) const { return wave; }
    // the next work chunk of this wave: one LDS atomic per wave (called with the wave converged), broadcast from its first lane
    __device__ __forceinline__ int next_chunk(int32_t *ctr, int, int, int) const {
        int c = 0;
        if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) c = atomicAdd(ctr, 1);
        return __builtin_amdgcn_readfirstlane(c);
    }
    // time a wave spends in one role of a phase: sum of the 100 MHz ticks in the low 40 bits, number of waves above
    __device__ __forceinline__ unsigned long long role_begin() const { return (PROF && prof) ? wall_clock64() : 0ull; }
    __device__ __forceinline__ void role_end(int id, unsigned long long start) const {
#ifdef RS_STUDY_SECTIONS
        return;
#endif
        if (PROF && prof && (threadIdx.x & 63) == 0 && (blockIdx.x & 15) == 0) atomicAdd(&prof[id], (wall_clock64() - start) + (1ull << 40));   // (every 16th environment: the sum stays below 2^40)
    }
    // the long code paths: called (production kernels) or inlined (the profiling kernel, whose section timers live inside them)
    template <class LT> __device__ __forceinline__ void plan_long(const KTab &T, const LT &L, const Grid &grid, const State &G, size_t eo, const KParams &P, int genv, int env, int t, int s) const {
#ifdef RS_CALL_LONG
        if constexpr (!PROF) { rs_fn_plan_long<CAP, REGS>(Ac, P.seed, P.sigma, genv, env, t, grid.tag, s); return; }
#endif
        phase_plan<true>(T, L, grid, G, eo, P, genv, t, s);
    }
    template <class LT> __device__ __forceinline__ void lc_decide(const KTab &T, const LT &L, const Grid &grid, const State &G, size_t eo, int env, int t, int s) const {
#ifdef RS_CALL_LONG
        if constexpr (!PROF) { rs_fn_lc_decide<CAP, REGS>(Ac, env, t, grid.tag, s); return; }
#endif
        lc_decide_and_flag(T, L, grid, G, eo, t, s);
    }
    template <class LT> __device__ __forceinline__ void move_long(const KTab &T, const LT &L, const Grid &gnew, const State &G, const KParams &P, int env, size_t eo, int t, bool last_tick,
                                                                  bool more, int s, int &active, int &halted, int &top) const {
#ifdef RS_CALL_LONG
        if constexpr (!PROF) { move_unpack(rs_fn_move_long<CAP, REGS>(Ac, P.out_mask, env, t, gnew.tag, (last_tick ? 1 : 0) | (more ? 2 : 0), s), active, halted, top); return; }
#endif
        phase_move<true>(T, L, gnew, G, P, env, eo, t, last_tick, more, s, active, halted, top);
    }
};
// Register budgets: 64 VGPRs (eight waves per SIMD: FOUR 512-thread workgroups per CU -- the default where the working memory of an
// environment fits four times, round 6) and 80 VGPRs (three 512-thread workgroups per CU; `_v128` is the same code under a third
// launch bound); CAP = the slot capacity as a compile-time constant (0: any).
template <int CAP, bool PROF, int REGS> __device__ __forceinline__ void rs_step_kernel_body(StepArgsPtr Ac, const KParams &P, const int32_t *__restrict__ actions) {
    if ((int)blockIdx.x >= P.n_envs) return;
    const StepArgs *A = (const StepArgs *)Ac;
#ifdef RS_STUDY_SECTIONS
    if (threadIdx.x == 0) g_sec_prof = P.prof;
#endif
    DevExec<PROF, CAP, REGS> ex{(int)blockDim.x, __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), PROF ? P.prof : nullptr, (PROF && P.prof) ? wall_clock64() : 0ull, Ac};
    rs_step_body<CAP>(ex, A->L, A->T, A->G, A->O, P, actions, (int)blockIdx.x);
}
template <int CAP>
__global__ void __launch_bounds__(1024, 8)
rs_step_kernel_v64(StepArgsPtr Ac, KParams P, const int32_t *__restrict__ actions) { rs_step_kernel_body<CAP, false, 64>(Ac, P, actions); }
template <int CAP>
__global__ void __launch_bounds__(768, 6)
rs_step_kernel_v80(StepArgsPtr Ac, KParams P, const int32_t *__restrict__ actions) { rs_step_kernel_body<CAP, false, 80>(Ac, P, actions); }
template <int CAP>
__global__ void __launch_bounds__(512, 4)
rs_step_kernel_v128(StepArgsPtr Ac, KParams P, const int32_t *__restrict__ actions) { rs_step_kernel_body<CAP, false, 128>(Ac, P, actions); }
// the profiling build (rs_phase_profile / RS_STUDY_SECTIONS): any capacity, 80 VGPRs
__global__ void __launch_bounds__(768, 6)
rs_step_kernel_prof(StepArgsPtr Ac, KParams P, const int32_t *__restrict__ actions) { rs_step_kernel_body<0, true, 80>(Ac, P, actions); }

// reset every environment: no vehicles, every backlog at its first trip, TLS programs freshly installed
// (Signal.__init__, traffic_signal.py:93-100)
__global__ void rs_reset_kernel(KTab T, State G, KParams P) {
    const int env = blockIdx.x;
    const int C = T.capacity, S = T.n_signals;
    const size_t eo = (size_t)env * C;
    for (int s = threadIdx.x; s < C; s += blockDim.x) {
        G.lane()[eo + s] = LANE_NONE; G.trip()[eo + s] = TRIP_NONE; G.owner()[eo + s] = OWNER_NONE;
        G.rwait()[eo + s] = 0; G.swait()[eo + s] = 0; G.cursor()[eo + s] = 0; G.depart()[eo + s] = 0; G.wtot()[eo + s] = 0;
        G.pos()[eo + s] = 0.0f; G.speed()[eo + s] = 0.0f; G.accel()[eo + s] = 0.0f; G.tloss()[eo + s] = 0.0f; G.sf()[eo + s] = 1.0f;
        G.coop(0)[eo + s] = COOP_NONE; G.coop(1)[eo + s] = COOP_NONE; G.cooplead(0)[eo + s] = COOP_NONE; G.cooplead(1)[eo + s] = COOP_NONE;
    }
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
        int ph, left;
        if (P.fixed_program) { ph = T.cold.fix_init_phase[s]; left = T.cold.fix_init_left[s]; }
        else { ph = T.cold.tls_init_phase[s]; left = T.cold.tls_dur[T.cold.tls_dur_off[s] + ph]; }
        G.tls[(env * S + s) * TLS_W + 0] = ph; G.tls[(env * S + s) * TLS_W + 1] = left; G.tls[(env * S + s) * TLS_W + 2] = 0; G.tls[(env * S + s) * TLS_W + 3] = 0;
    }
    for (int d = threadIdx.x; d < T.n_dep; d += blockDim.x) G.dep_next[(size_t)env * T.n_dep + d] = T.cold.dep_first[d];
    for (int i = threadIdx.x; i < (C + 31) / 32; i += blockDim.x) G.mail[(size_t)env * ((C + 31) / 32) + i] = 0u;
    if (threadIdx.x < 4) G.env[env * 4 + threadIdx.x] = 0;
    if (threadIdx.x < ST_N) G.stats[(size_t)env * ST_N + threadIdx.x] = 0;
    if (G.trip_log)
        for (int i = threadIdx.x; i < T.n_trips * 4; i += blockDim.x) G.trip_log[(size_t)env * T.n_trips * 4 + i] = 0;
}

// fresh Signal objects on the running simulation (rs_reinit_signals)
__global__ void rs_reinit_kernel(KTab T, State G, KParams P) {
    const int env = blockIdx.x;
    const int C = T.capacity, S = T.n_signals;
    const size_t eo = (size_t)env * C;
    for (int s = threadIdx.x; s < C; s += blockDim.x) { G.owner()[eo + s] = OWNER_NONE; G.rwait()[eo + s] = 0; }
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
        if (!P.fixed_program) G.tls[(env * S + s) * TLS_W + 1] = T.cold.tls_dur[T.cold.tls_dur_off[s] + G.tls[(env * S + s) * TLS_W + 0]];
        G.tls[(env * S + s) * TLS_W + 2] = 0; G.tls[(env * S + s) * TLS_W + 3] = 0;
    }
}

// ---- static agents
// STOCHASTIC (agents/stochastic.py:17-18): uniform green index per (env, signal, step)
__global__ void rs_act_random_kernel(KTab T, KParams P, uint32_t step_key, int32_t *actions) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int S = T.n_signals;
    if (i >= P.n_envs * S) return;
    const int env = i / S, s = i - env * S;
    const uint32_t h = d_hash(P.seed ^ 0xA5A5A5A5u, (uint32_t)(P.env_base + env), (uint32_t)s, step_key, 7u);
    actions[i] = (int32_t)(h % (uint32_t)T.cold.tls_ngreen[s]);
}
// MAXWAVE / MAXPRESSURE (agents/maxwave.py:18-38, maxpressure.py:13-18): first maximum over the valid
// phase pairs (in the reference's iteration order) of obs[pair0] + obs[pair1]
__global__ void rs_act_maxwave_kernel(KTab T, KParams P, const int32_t *pairs, int n_pairs, const int32_t *valid, const int32_t *order,
                                      int use_pressure, const int32_t *mplight, const int32_t *wave, int32_t *actions) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int S = T.n_signals;
    if (i >= P.n_envs * S) return;
    const int s = i % S;
    const int32_t *obs = use_pressure ? mplight + (size_t)i * 13 + 1 : wave + (size_t)i * 12;
    bool have = false;
    int best = 0, best_act = 0;
    for (int j = 0; j < n_pairs; ++j) {
        const int p = order[s * n_pairs + j];     // the reference walks valid_acts in dict order; ties keep the first
        if (p < 0) break;
        const int act = valid[s * n_pairs + p];
        if (act < 0) continue;
        const int press = obs[pairs[p * 2]] + obs[pairs[p * 2 + 1]];
        if (!have || press > best) { have = true; best = press; best_act = act; }
    }
    actions[i] = best_act;
}

// ------------------------------------------------------------------------------------------------ host side
struct rs_sim {
    int device = 0;
    int n_envs = 0, env_base = 0, block = 256;
    int ratio = 1;                  // rs_params.step_ratio: simulation ticks per step_sim() call
    size_t lds = 0;
    KTab K{};
    StepArgs *args = nullptr;      // device copy of {K, G, O}
    int use_v128 = 0;
    State G{};
    Out O{};
    KParams P{};
    uint32_t out_mask = OUT_ALL;        // rs_set_outputs
    int32_t *actions = nullptr;
    int32_t *pairs = nullptr, *valid = nullptr, *order = nullptr;
    unsigned long long *prof = nullptr;
    int n_pairs = 0;
    hipStream_t stream = nullptr;
    hipStream_t last = nullptr;         // stream of the most recent launch: what the synchronous calls wait for
    std::vector<void *> allocs;
    std::vector<int32_t> tls_ngreen;
    struct Buf { void *ptr; int64_t shape[4]; int ndim; int dtype; size_t bytes; };
    Buf bufs[RS_BUF_COUNT]{};
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t ev_used = 0;
    std::string err;
};

static thread_local std::string g_create_err;

#define HIPCHK(h, call)                                                                            \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                          \
            (void)hipGetLastError();    /* the error is reported here: do not leave it for the next hipGetLastError() */ \
            return RS_EHIP;                                                                        \
        }                                                                                          \
    } while (0)

template <typename Tp>
static int dev_alloc(rs_sim *h, Tp **p, size_t count, bool zero = true) {
    void *d = nullptr;
    size_t bytes = (count ? count : 1) * sizeof(Tp);
    hipError_t e = hipMalloc(&d, bytes);
    if (e != hipSuccess) { h->err = std::string("hipMalloc: ") + hipGetErrorString(e); return RS_ENOMEM; }
    if (zero) (void)hipMemset(d, 0, bytes);
    h->allocs.push_back(d);
    *p = (Tp *)d;
    return RS_OK;
}
template <typename Tp>
static int dev_upload(rs_sim *h, const Tp **dst, const Tp *src, size_t count) {
    Tp *d = nullptr;
    int rc = dev_alloc(h, &d, count, false);
    if (rc) return rc;
    if (count) HIPCHK(h, hipMemcpy(d, src, count * sizeof(Tp), hipMemcpyHostToDevice));
    *dst = d;
    return RS_OK;
}

static const size_t kDtypeSize[] = {4, 4, 2, 1, 2, 8, 4};
static void set_buf(rs_sim *h, int which, void *ptr, int dtype, int ndim, int64_t a, int64_t b = 1, int64_t c = 1, int64_t d = 1) {
    auto &B = h->bufs[which];
    B.ptr = ptr; B.dtype = dtype; B.ndim = ndim;
    B.shape[0] = a; B.shape[1] = b; B.shape[2] = c; B.shape[3] = d;
    B.bytes = (size_t)(a * b * c * d) * kDtypeSize[dtype];
}

typedef void (*step_kernel_fn)(StepArgsPtr, KParams, const int32_t *);
static const int kStepCaps[] = {0, 128, 256, 512, 768, 896, 1024};
// regs: 0 = 64 VGPRs (blocks up to 1024 threads), 1 = 128 VGPRs, 2 = 80 VGPRs (blocks up to 512 threads)
#define RS_PICK(cap_) (regs == 1 ? rs_step_kernel_v128<cap_> : (regs == 2 ? rs_step_kernel_v80<cap_> : rs_step_kernel_v64<cap_>))
static step_kernel_fn step_kernel_for(int regs, int capacity) {
#ifdef RS_ONE_CAP       // study builds (seconds instead of minutes to compile): one capacity, the 64- and the 80-VGPR kernel only
    (void)capacity;
    return regs == 0 ? rs_step_kernel_v64<RS_ONE_CAP> : rs_step_kernel_v80<RS_ONE_CAP>;
#else
    switch (capacity) {
        case 128: return RS_PICK(128);
        case 256: return RS_PICK(256);
        case 512: return RS_PICK(512);
        case 768: return RS_PICK(768);
        case 896: return RS_PICK(896);
        case 1024: return RS_PICK(1024);
        default: return RS_PICK(0);
    }
#endif
}

// every synchronous entry point waits for the handle's own stream AND the caller stream of the last launch
static hipError_t wait_idle(rs_sim *h) {
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess && h->last && h->last != h->stream) e = hipStreamSynchronize(h->last);
    return e;
}

// The workgroup shape rs_create picks for block_threads = 0 (include/resco_sim.h).  Base: one thread per TWO slots (the capacity is the
// episode's peak, about twice the typical occupancy), at most 512 threads, 80 VGPRs -- at large batches the fewest waves per
// environment win (cologne1, 128 slots: 64 threads 13.3 M env-steps/s against 12.5 M with 128 at 16 384 environments).  A SMALL batch
// leaves the chip empty at that shape -- 1024 environments x one wave are 4 waves per CU -- and a phase of the tick is the latency of
// its chunks one after the other on that wave: as long as the resident-wave budget of the device (7 waves per SIMD with the 80-VGPR
// build) holds every environment at once, the environment gets more waves, up to one per chunk of a phase (capacity / 64 slots chunks
// + one list chunk).  Measured on one MI355X, random policy (profiles/r06_block_sweep.txt): cologne1 x 1024 2.65 -> 4.94 M with 256
// threads, cologne8 x 2048 6.04 -> 6.88 M with 192, and at >= 4096 environments the base shape again.
extern "C" int32_t rs_default_block(int32_t capacity, int32_t n_envs_on_device, int32_t device_id) {
    const int C = capacity;
    if (C < 64 || n_envs_on_device <= 0) return 0;
    const int base = C >= 768 ? 8 : ((C / 2) + 63) / 64;            // waves
    int waves = base;
    if (C < 768) {          // (the large scenarios run three or four 512-thread workgroups per CU: more threads would lose one)
        int cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        else (void)hipGetLastError();
        const long budget = (long)cus * 4 * 7;                      // resident waves of the 80-VGPR build (what these shapes run with)
        const int fit = (int)(budget / n_envs_on_device), most = C / 64 + 1;
        waves = fit < most ? fit : most;
        if (waves < base) waves = base;
    }
    return -(20000 + 64 * waves);
}

extern "C" int rs_create(const rs_scenario *sc, const rs_params *p, int32_t n_envs, int32_t env_base, int32_t device_id,
                         int32_t block_threads, rs_handle *out) {
    if (!sc || !p || !out || n_envs <= 0) { g_create_err = "rs_create: bad argument"; return RS_EINVAL; }
    rs_sim *h = new (std::nothrow) rs_sim();
    if (!h) return RS_ENOMEM;
    auto fail = [&](int rc) { g_create_err = h->err; (void)hipGetLastError(); rs_destroy(h); return rc; };
    (void)hipGetLastError();        // a stale error of this thread (another library's, an earlier failed call) is not ours
    h->device = device_id; h->n_envs = n_envs; h->env_base = env_base;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { h->err = "no HIP device visible (this library has no CPU fallback)"; return fail(RS_EHIP); }
    if (hipSetDevice(device_id) != hipSuccess) { h->err = "hipSetDevice failed"; return fail(RS_EHIP); }
    if (sc->step_length <= 0 || sc->yellow_length < 0 || sc->yellow_length >= sc->step_length) {
        h->err = "need 0 <= yellow_length < step_length"; return fail(RS_EINVAL);
    }
    const int C = sc->capacity;
    if (C < 64 || (C % 64) || C > 1984) { h->err = "capacity must be a multiple of 64 in [64, 1984]"; return fail(RS_ELIMIT); }
    if (sc->kmax < 1 || sc->kmax > 16) { h->err = "kmax (lanes per edge) must be in [1, 16]"; return fail(RS_ELIMIT); }
    PackedTables PT;
    {   // grid cell length: build once to learn the sizes that do not depend on it, choose, build for good
        PackedTables probe;
        if (!probe.build(sc)) { h->err = probe.err; return fail(RS_ELIMIT); }
        if (!PT.build(sc, pick_cell_len(sc, probe.n_arr, probe.n_dep, probe.tls_maxl))) { h->err = PT.err; return fail(RS_ELIMIT); }
    }
    h->tls_ngreen.assign(sc->tls_ngreen, sc->tls_ngreen + sc->n_signals);
    int rc;
    KTab &K = h->K;
    {
        KCold &cold = K.cold;
#define UP(dst, type, src, count) if ((rc = dev_upload<type>(h, &dst, src, (size_t)(count)))) return fail(rc);
        UP(K.lanes_, LaneRec, PT.lanes.data(), PT.lanes.size()) UP(K.links_, LinkRec, PT.links.data(), PT.links.size())
        UP(K.foes_, FoeRec, PT.foes.data(), PT.foes.size()) UP(K.rsteps_, RStep, PT.rsteps.data(), PT.rsteps.size())
        UP(K.routes_, RouteRec, PT.routes.data(), PT.routes.size()) UP(K.next_link_, uint16_t, PT.next_link.data(), PT.next_link.size())
        UP(K.trip_route_, uint16_t, PT.trip_route.data(), PT.trip_route.size()) UP(K.trip_vtype_, uint8_t, PT.trip_vtype.data(), PT.trip_vtype.size())
        UP(K.route_cont_, float, PT.route_cont.data(), PT.route_cont.size()) UP(K.notbest_, uint16_t, PT.notbest.data(), PT.notbest.size())
        UP(cold.trip_depart, int32_t, sc->trip_depart, sc->n_trips) UP(cold.trip_next, uint16_t, PT.trip_next.data(), PT.trip_next.size())
        UP(cold.dep_lane, uint16_t, PT.dep_lane.data(), PT.dep_lane.size()) UP(cold.dep_info, DepInfo, PT.dep_info.data(), PT.dep_info.size()) UP(cold.dep_first, uint16_t, PT.dep_first.data(), PT.dep_first.size())
        UP(cold.vtype_params, float, sc->vtype_params, sc->n_vtypes * VT_COLS)
        UP(cold.tls8, uint8_t, PT.tls8.data(), PT.tls8.size()) UP(cold.fix8, uint8_t, PT.fix8.data(), PT.fix8.size())
        UP(cold.tls_nphase, int32_t, sc->tls_nphase, sc->n_signals) UP(cold.tls_ngreen, int32_t, sc->tls_ngreen, sc->n_signals)
        UP(cold.tls_nlinks, int32_t, sc->tls_nlinks, sc->n_signals) UP(cold.tls_state_off, int32_t, PT.tls_off_p.data(), sc->n_signals)
        UP(cold.tls_dur_off, int32_t, sc->tls_dur_off, sc->n_signals) UP(cold.tls_yel_off, int32_t, sc->tls_yel_off, sc->n_signals)
        UP(cold.tls_dur, int32_t, sc->tls_dur, sc->n_tls_dur) UP(cold.tls_yellow, int32_t, sc->tls_yellow, sc->n_tls_yellow)
        UP(cold.tls_init_phase, int32_t, sc->tls_init_phase, sc->n_signals)
        UP(cold.fix_nphase, int32_t, sc->fix_nphase, sc->n_signals) UP(cold.fix_state_off, int32_t, PT.fix_off_p.data(), sc->n_signals)
        UP(cold.fix_dur_off, int32_t, sc->fix_dur_off, sc->n_signals) UP(cold.fix_dur, int32_t, sc->fix_dur, sc->n_fix_dur)
        UP(cold.fix_init_phase, int32_t, sc->fix_init_phase, sc->n_signals) UP(cold.fix_init_left, int32_t, sc->fix_init_left, sc->n_signals)
        UP(cold.lane_obs, int16_t, PT.lane_obs16.data(), PT.lane_obs16.size()) UP(cold.obs_sig, int32_t, PT.obs_sig.data(), PT.obs_sig.size())
        UP(cold.sig_obs_start, int32_t, sc->sig_obs_start, sc->n_signals + 1)
        UP(cold.mv_in_start, int32_t, sc->mv_in_start, sc->n_signals * 12 + 1) UP(cold.mv_in_idx, int32_t, sc->mv_in_idx, sc->n_mv_in)
        UP(cold.mv_out_start, int32_t, sc->mv_out_start, sc->n_signals * 12 + 1) UP(cold.mv_out_idx, int32_t, sc->mv_out_idx, sc->n_mv_out)
        UP(cold.pr_out_start, int32_t, sc->pr_out_start, sc->n_signals + 1) UP(cold.pr_out_idx, int32_t, sc->pr_out_idx, sc->n_pr_out)
        UP(cold.trips_cum, int32_t, sc->trips_cum, sc->horizon + 2)
#undef UP
        K.maxlen = PT.maxlen; K.occ_unit = PT.occ_unit;
        K.n_trips = sc->n_trips; K.tls_maxl = PT.tls_maxl; K.kmax = sc->kmax;
        K.n_lanes = sc->n_lanes; K.n_cells = PT.n_cells; K.n_signals = sc->n_signals; K.n_obs = sc->n_obs; K.n_vtypes = sc->n_vtypes;
        h->ratio = p->step_ratio > 1 ? p->step_ratio : 1;
        // the kernel counts ticks: Signal.set_phase comes after yellow_length x step_ratio of them (multi_signal.py:102-105, 175-180);
        // step_length stays what Signal.observe adds to a waiting time (traffic_signal.py:196)
        K.horizon = sc->horizon; K.capacity = C; K.step_length = sc->step_length; K.yellow_length = sc->yellow_length * h->ratio; K.lmax = PT.lmax;
        K.n_arr = PT.n_arr; K.n_dep = PT.n_dep;
    }
    const int lmax = PT.lmax;

    h->P.seed = p->seed; h->P.env_base = env_base; h->P.max_distance = p->max_distance; h->P.sigma = p->sigma;
    h->P.speed_dev = p->speed_dev; h->P.fixed_program = p->fixed_program; h->P.tls_expiry = p->tls_hold == 0; h->P.n_envs = n_envs;

    const size_t N = (size_t)n_envs, NC = N * C, S = (size_t)sc->n_signals, NO = (size_t)sc->n_obs;
    State &G = h->G;
    Out &O = h->O;
    {
        char *slab = nullptr, *outb = nullptr;
        G.nc = NC;
        O.n = n_envs; O.o = sc->n_obs; O.s = sc->n_signals; O.lm = lmax;
        if ((rc = dev_alloc(h, &slab, State::bytes(NC))) || (rc = dev_alloc(h, &outb, O.bytes())) ||
            (rc = dev_alloc(h, &G.env, N * 4)) || (rc = dev_alloc(h, &G.tls, N * S * TLS_W)) || (rc = dev_alloc(h, &G.stats, N * ST_N)) ||
            (rc = dev_alloc(h, &G.dep_next, N * (size_t)h->K.n_dep)) || (rc = dev_alloc(h, &G.mail, N * (size_t)((C + 31) / 32))) ||
            (rc = dev_alloc(h, &h->actions, N * S)))
            return fail(rc);
        G.base = slab; O.base = outb;
        (void)NO;
    }
    const int64_t n = n_envs, c = C, s = sc->n_signals, o = sc->n_obs;
    set_buf(h, RS_BUF_LANE_AGG, O.lane_agg(), RS_F32, 3, n, o, 5);
    set_buf(h, RS_BUF_DRQ_NORM, O.drq_norm(), RS_F32, 3, n, o, 5);
    set_buf(h, RS_BUF_PHASE, O.phase(), RS_I32, 2, n, s);
    set_buf(h, RS_BUF_MPLIGHT, O.mplight(), RS_I32, 3, n, s, 13);
    set_buf(h, RS_BUF_WAVE, O.wave(), RS_I32, 3, n, s, 12);
    set_buf(h, RS_BUF_WAIT, O.wait(), RS_F32, 2, n, s);
    set_buf(h, RS_BUF_WAIT_NORM, O.wait_norm(), RS_F32, 2, n, s);
    set_buf(h, RS_BUF_PRESSURE, O.pressure(), RS_I32, 2, n, s);
    set_buf(h, RS_BUF_QUEUE_SUM, O.queue_sum(), RS_I32, 2, n, s);
    set_buf(h, RS_BUF_QUEUE_MAX, O.queue_max(), RS_I32, 2, n, s);
    set_buf(h, RS_BUF_ACTIONS, h->actions, RS_I32, 2, n, s);
    set_buf(h, RS_BUF_ENV, G.env, RS_I32, 2, n, 4);
    set_buf(h, RS_BUF_TLS, G.tls, RS_I32, 3, n, s, TLS_W);
    set_buf(h, RS_BUF_VEH_POS, G.pos(), RS_F32, 2, n, c);
    set_buf(h, RS_BUF_VEH_SPEED, G.speed(), RS_F32, 2, n, c);
    set_buf(h, RS_BUF_VEH_ACCEL, G.accel(), RS_F32, 2, n, c);
    set_buf(h, RS_BUF_VEH_TLOSS, G.tloss(), RS_F32, 2, n, c);
    set_buf(h, RS_BUF_VEH_LANE, G.lane(), RS_U16, 2, n, c);
    set_buf(h, RS_BUF_VEH_TRIP, G.trip(), RS_U16, 2, n, c);
    set_buf(h, RS_BUF_VEH_CURSOR, G.cursor(), RS_U16, 2, n, c);
    set_buf(h, RS_BUF_VEH_SWAIT, G.swait(), RS_U16, 2, n, c);
    set_buf(h, RS_BUF_VEH_RWAIT, G.rwait(), RS_U16, 2, n, c);
    set_buf(h, RS_BUF_VEH_DEPART, G.depart(), RS_U16, 2, n, c);
    set_buf(h, RS_BUF_VEH_OWNER, G.owner(), RS_U8, 2, n, c);
    set_buf(h, RS_BUF_STATS, G.stats, RS_I64, 2, n, ST_N);
    set_buf(h, RS_BUF_DRQ_NORM_F16, O.drq_f16(), RS_F16, 4, n, s, lmax, 5);
    set_buf(h, RS_BUF_VEH_SF, G.sf(), RS_F32, 2, n, c);
    set_buf(h, RS_BUF_VEH_WTOT, G.wtot(), RS_U16, 2, n, c);
    G.trip_log = nullptr;
    if (p->trip_log && (rc = dev_alloc(h, &G.trip_log, N * (size_t)sc->n_trips * 4))) return fail(rc);
    set_buf(h, RS_BUF_TRIP_LOG, G.trip_log, RS_I32, 3, n, p->trip_log ? sc->n_trips : 0, 4);
    set_buf(h, RS_BUF_DEP_NEXT, G.dep_next, RS_U16, 2, n, h->K.n_dep);
    set_buf(h, RS_BUF_VEH_COOP, G.coop(0), RS_U32, 2, n, c);
    set_buf(h, RS_BUF_VEH_COOPLEAD, G.cooplead(0), RS_U32, 2, n, c);
    set_buf(h, RS_BUF_ARRIVALS, O.arrivals(), RS_I32, 2, n, s);
    set_buf(h, RS_BUF_DEPARTURES, O.departures(), RS_I32, 2, n, s);
    set_buf(h, RS_BUF_MPLIGHT_FULL, O.mplight_full(), RS_F32, 3, n, s, 49);
    set_buf(h, RS_BUF_LANE_ARRIVALS, O.lane_arr(), RS_I32, 2, n, sc->n_obs);
    set_buf(h, RS_BUF_VEH_COOP_ODD, G.coop(1), RS_U32, 2, n, c); set_buf(h, RS_BUF_VEH_COOPLEAD_ODD, G.cooplead(1), RS_U32, 2, n, c);
    set_buf(h, RS_BUF_VEH_MAIL, G.mail, RS_U32, 2, n, (C + 31) / 32);

    h->lds = lds_carve(nullptr, C, h->K.n_cells, h->K.n_arr, h->K.n_dep, sc->n_obs, sc->n_signals, sc->n_vtypes, h->K.tls_maxl);
    if (const char *pad = getenv("RESCO_STUDY_LDS_PAD")) h->lds += (size_t)atoi(pad);      // study knob: unused bytes, to hold the residency fixed in an A/B
    if (h->lds > 160 * 1024) { h->err = "scenario needs more than 160 KiB of LDS per environment"; return fail(RS_ELIMIT); }
    // block_threads: 0 = one thread per slot (at most 1024); a negative value selects the 128-VGPR build with |value|
    // threads (<= 512), -(10000 + threads) the 80-VGPR build -- tuning knobs, see DESIGN.md
    if (block_threads == 0) block_threads = rs_default_block(C, n_envs, device_id);
    if (block_threads <= -20000) {
        // the shape rs_default_block proposes: the register budget follows from what fits a CU.  Where the working memory lets FOUR
        // 512-thread workgroups share a CU they need eight waves per SIMD, i.e. the 64-VGPR build (ingolstadt21 with 896 slots:
        // 40 768 B; +14 % env-steps/s over three workgroups of the 80-VGPR build, profiles/r06_ab_occupancy.txt); else 80 VGPRs
        block_threads = -block_threads - 20000;
        h->use_v128 = (block_threads == 512 && h->lds <= RS_LDS_4WG_LIMIT) ? 0 : 2;
    } else if (block_threads < 0) { h->use_v128 = 1; block_threads = -block_threads; if (block_threads >= 10000) { h->use_v128 = 2; block_threads -= 10000; } }
    if (block_threads % 64 || block_threads > (h->use_v128 == 1 ? 512 : (h->use_v128 == 2 ? 768 : 1024)) || block_threads < 64) {
        h->err = "block_threads must be a multiple of 64 in [64, 1024] ([64, 768] for the 80-VGPR build, [64, 512] for the 128-VGPR build)";
        return fail(RS_EINVAL);
    }
    h->block = block_threads;
    {
        // the dynamic-LDS ceiling is an attribute of the kernel (per device), not of a launch: only ever raise it,
        // or a handle created earlier for a larger scenario could no longer launch
        static std::mutex mu;
        static size_t max_lds[64] = {0};
        std::lock_guard<std::mutex> lock(mu);
        size_t &cur = max_lds[device_id & 63];
        if (h->lds > cur) {
            for (int v = 0; v < 3; ++v)
                for (int cp : kStepCaps)        // every instantiation: the ceiling is per kernel function
                    if (hipFuncSetAttribute((const void *)step_kernel_for(v, cp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds) != hipSuccess) {
                        h->err = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed"; return fail(RS_EHIP);
                    }
            if (hipFuncSetAttribute((const void *)rs_step_kernel_prof, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds) != hipSuccess) {
                h->err = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed"; return fail(RS_EHIP);
            }
            cur = h->lds;
        }
    }
    {
        StepArgs sa{h->K, h->G, h->O, Lds{}};
        lds_carve(&sa.L, C, h->K.n_cells, h->K.n_arr, h->K.n_dep, sc->n_obs, sc->n_signals, sc->n_vtypes, h->K.tls_maxl);
        if (!lds_fix_matches(sa.L, C)) { h->err = "the layout of the working memory does not match the kernel's literals (lds_carve / LdsFix)"; return fail(RS_EINVAL); }
        sa.L.cell_inv = PT.cell_inv;
        if ((rc = dev_alloc(h, &h->args, 1, false))) return fail(rc);
        if (hipMemcpy(h->args, &sa, sizeof(sa), hipMemcpyHostToDevice) != hipSuccess) { h->err = "hipMemcpy(StepArgs) failed"; return fail(RS_EHIP); }
    }
    {   // The handle's stream gets a hardware queue of its OWN.  HIP multiplexes plain streams over GPU_MAX_HW_QUEUES (4) hardware
        // queues, least-used first, and two streams that share one run their kernels one after the other: pipes (several handles
        // per GPU whose launches are meant to overlap) lost a third of their rate whenever two of them met on a queue
        // (profiles/r05_pipes_group.txt).  A stream created with a CU mask is never multiplexed; the mask enables every CU.
        hipDeviceProp_t prop;
        std::vector<uint32_t> mask;
        if (getenv("RESCO_PLAIN_STREAMS") == nullptr && hipGetDeviceProperties(&prop, device_id) == hipSuccess)
            mask.assign((size_t)(prop.multiProcessorCount + 31) / 32, 0xFFFFFFFFu);
        if (mask.empty() || hipExtStreamCreateWithCUMask(&h->stream, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
            (void)hipGetLastError();
            if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { h->err = "hipStreamCreate failed"; return fail(RS_EHIP); }
        }
    }
    *out = h;
    int r2 = rs_reset(h, nullptr);
    if (r2) { g_create_err = h->err; *out = nullptr; rs_destroy(h); return r2; }
    return RS_OK;
}

extern "C" void rs_destroy(rs_handle h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) { (void)wait_idle(h); (void)hipStreamDestroy(h->stream); }
    for (auto &e : h->events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (void *p : h->allocs) (void)hipFree(p);
    delete h;
}

extern "C" const char *rs_last_error(rs_handle h) { return h ? h->err.c_str() : g_create_err.c_str(); }

static int launch_step(rs_sim *h, hipStream_t st, int n_ticks, int do_fsm, int do_observe = 1) {
    KParams P = h->P;
    P.n_ticks = n_ticks; P.do_fsm = do_fsm; P.do_observe = do_observe; P.out_mask = h->out_mask; P.prof = h->prof;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->timing) {
        if (h->ev_used == h->events.size()) {
            hipEvent_t a, b;
            HIPCHK(h, hipEventCreate(&a));
            HIPCHK(h, hipEventCreate(&b));
            h->events.emplace_back(a, b);
        }
        e0 = h->events[h->ev_used].first; e1 = h->events[h->ev_used].second;
        h->ev_used += 1;
        HIPCHK(h, hipEventRecord(e0, st));
    }
    // (the in-kernel timers live in a kernel of their own: any capacity, 80 VGPRs, at most 768 threads)
    const step_kernel_fn fn = (h->prof && h->block <= 768) ? (step_kernel_fn)rs_step_kernel_prof : step_kernel_for(h->use_v128, h->K.capacity);
    hipLaunchKernelGGL(fn, dim3(h->n_envs), dim3(h->block), h->lds, st, (StepArgsPtr)h->args, P, (const int32_t *)h->actions);
    HIPCHK(h, hipGetLastError());
    if (h->timing) HIPCHK(h, hipEventRecord(e1, st));
    return RS_OK;
}

extern "C" int rs_reset(rs_handle h, void *stream) {
    if (!h) return RS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    h->last = st;
    hipLaunchKernelGGL(rs_reset_kernel, dim3(h->n_envs), dim3(256), 0, st, h->K, h->G, h->P);
    HIPCHK(h, hipGetLastError());
    bool tm = h->timing;
    h->timing = false;
    int rc = launch_step(h, st, 0, 0);
    h->timing = tm;
    return rc;
}

extern "C" int rs_reinit_signals(rs_handle h, void *stream) {
    if (!h) return RS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    h->last = st;
    hipLaunchKernelGGL(rs_reinit_kernel, dim3(h->n_envs), dim3(256), 0, st, h->K, h->G, h->P);
    HIPCHK(h, hipGetLastError());
    bool tm = h->timing;
    h->timing = false;
    int rc = launch_step(h, st, 0, 0);      // the first observe of the new Signal objects
    h->timing = tm;
    return rc;
}

extern "C" int rs_step(rs_handle h, const int32_t *actions, int32_t actions_on_device, void *stream) {
    if (!h) return RS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    h->last = st;
    if (actions) {
        size_t bytes = (size_t)h->n_envs * h->K.n_signals * sizeof(int32_t);
        HIPCHK(h, hipMemcpyAsync(h->actions, actions, bytes, actions_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
        // a pageable host source may be read after the call returns: make the caller's buffer reusable
        if (!actions_on_device) HIPCHK(h, hipStreamSynchronize(st));
    }
    return launch_step(h, st, h->K.step_length * h->ratio, 1);
}

extern "C" int rs_ticks(rs_handle h, int32_t n_ticks, void *stream) {
    if (!h || n_ticks < 0) return RS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    h->last = st;
    return launch_step(h, st, n_ticks, 0);
}

extern "C" int rs_step_sim(rs_handle h, int32_t n_ticks, void *stream) {
    if (!h || n_ticks < 0) return RS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    h->last = st;
    return launch_step(h, st, n_ticks, 0, 0);
}

// which output buffers the observe of the following launches writes (bit b = buffer id b); the others keep their contents
extern "C" int rs_set_outputs(rs_handle h, uint64_t buffer_mask) {
    if (!h) return RS_EINVAL;
    uint32_t m = 0;
    if (buffer_mask & (1ull << RS_BUF_LANE_AGG)) m |= OUT_LANE_AGG;
    if (buffer_mask & (1ull << RS_BUF_DRQ_NORM)) m |= OUT_DRQ_NORM;
    if (buffer_mask & (1ull << RS_BUF_DRQ_NORM_F16)) m |= OUT_DRQ_F16;
    if (buffer_mask & (1ull << RS_BUF_LANE_ARRIVALS)) m |= OUT_LANE_ARR;
    if (buffer_mask & (1ull << RS_BUF_MPLIGHT)) m |= OUT_MPLIGHT;
    if (buffer_mask & (1ull << RS_BUF_WAVE)) m |= OUT_WAVE;
    if (buffer_mask & (1ull << RS_BUF_MPLIGHT_FULL)) m |= OUT_MPLIGHT_FULL;
    if (buffer_mask & (1ull << RS_BUF_VEH_ACCEL)) m |= OUT_VEH_ACCEL;
    h->out_mask = m;
    return RS_OK;
}

extern "C" int rs_sync(rs_handle h) {
    if (!h) return RS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, wait_idle(h));
    return RS_OK;
}

extern "C" int rs_act_random(rs_handle h, uint32_t step_key, void *stream) {
    if (!h) return RS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    h->last = st;
    int total = h->n_envs * h->K.n_signals;
    hipLaunchKernelGGL(rs_act_random_kernel, dim3((total + 255) / 256), dim3(256), 0, st, h->K, h->P, step_key, h->actions);
    HIPCHK(h, hipGetLastError());
    return RS_OK;
}

extern "C" int rs_act_maxwave(rs_handle h, const int32_t *phase_pairs, int32_t n_pairs, const int32_t *valid,
                              const int32_t *order, int32_t use_pressure, void *stream) {
    if (!h || n_pairs <= 0) return RS_EINVAL;
    // the agent reads the states.mplight / states.wave rows of the last observe: refuse when rs_set_outputs switched them off
    // (the buffer would hold stale rows, or zeros if it was never written)
    if (!(h->out_mask & (use_pressure ? OUT_MPLIGHT : OUT_WAVE))) {
        h->err = use_pressure ? "rs_act_maxwave(use_pressure=1) reads RS_BUF_MPLIGHT, which rs_set_outputs has switched off"
                              : "rs_act_maxwave(use_pressure=0) reads RS_BUF_WAVE, which rs_set_outputs has switched off";
        return RS_EINVAL;
    }
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    h->last = st;
    if (!h->pairs) {
        if (!phase_pairs || !valid || !order) { h->err = "rs_act_maxwave: tables required on first use"; return RS_EINVAL; }
        int rc;
        if ((rc = dev_alloc(h, &h->pairs, (size_t)n_pairs * 2, false)) || (rc = dev_alloc(h, &h->valid, (size_t)h->K.n_signals * n_pairs, false)) ||
            (rc = dev_alloc(h, &h->order, (size_t)h->K.n_signals * n_pairs, false))) return rc;
        HIPCHK(h, hipMemcpy(h->order, order, (size_t)h->K.n_signals * n_pairs * 4, hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(h->pairs, phase_pairs, (size_t)n_pairs * 2 * 4, hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(h->valid, valid, (size_t)h->K.n_signals * n_pairs * 4, hipMemcpyHostToDevice));
        h->n_pairs = n_pairs;
    }
    int total = h->n_envs * h->K.n_signals;
    hipLaunchKernelGGL(rs_act_maxwave_kernel, dim3((total + 255) / 256), dim3(256), 0, st, h->K, h->P, (const int32_t *)h->pairs,
                       h->n_pairs, (const int32_t *)h->valid, (const int32_t *)h->order, (int)use_pressure, (const int32_t *)h->O.mplight(),
                       (const int32_t *)h->O.wave(), h->actions);
    HIPCHK(h, hipGetLastError());
    return RS_OK;
}

extern "C" int rs_get_buffer(rs_handle h, int32_t which, void **dev_ptr, int64_t shape[4], int32_t *ndim, int32_t *dtype) {
    if (!h || which < 0 || which >= RS_BUF_COUNT) return RS_EINVAL;
    auto &B = h->bufs[which];
    if (dev_ptr) *dev_ptr = B.ptr;
    if (shape) for (int i = 0; i < 4; ++i) shape[i] = B.shape[i];
    if (ndim) *ndim = B.ndim;
    if (dtype) *dtype = B.dtype;
    return RS_OK;
}

extern "C" int rs_read_buffer(rs_handle h, int32_t which, void *host_dst, int64_t nbytes) {
    if (!h || which < 0 || which >= RS_BUF_COUNT || !host_dst) return RS_EINVAL;
    auto &B = h->bufs[which];
    if ((size_t)nbytes != B.bytes) { h->err = "rs_read_buffer: size mismatch"; return RS_EINVAL; }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, wait_idle(h));
    HIPCHK(h, hipMemcpy(host_dst, B.ptr, B.bytes, hipMemcpyDeviceToHost));
    return RS_OK;
}

extern "C" int rs_stats(rs_handle h, int64_t *host_out) {
    if (!h) return RS_EINVAL;
    return rs_read_buffer(h, RS_BUF_STATS, host_out, (int64_t)h->n_envs * ST_N * 8);
}

struct Snapshot { std::vector<void *> ptrs; uint32_t seed = 0; };     // the seed: the speed factors are recomputed from (seed, env, trip) at every load
// state AND the observation buffers: re-running observe would advance Signal.waiting_times
static const int kSnapBufs[] = {RS_BUF_LANE_AGG, RS_BUF_DRQ_NORM, RS_BUF_PHASE, RS_BUF_MPLIGHT, RS_BUF_WAVE, RS_BUF_WAIT,
                                RS_BUF_WAIT_NORM, RS_BUF_PRESSURE, RS_BUF_QUEUE_SUM, RS_BUF_QUEUE_MAX, RS_BUF_DRQ_NORM_F16,
                                RS_BUF_ENV, RS_BUF_TLS, RS_BUF_VEH_POS, RS_BUF_VEH_SPEED, RS_BUF_VEH_ACCEL, RS_BUF_VEH_TLOSS,
                                RS_BUF_VEH_LANE, RS_BUF_VEH_TRIP, RS_BUF_VEH_CURSOR, RS_BUF_VEH_SWAIT, RS_BUF_VEH_RWAIT,
                                RS_BUF_VEH_DEPART, RS_BUF_VEH_OWNER, RS_BUF_VEH_SF, RS_BUF_VEH_WTOT, RS_BUF_TRIP_LOG, RS_BUF_STATS,
                                RS_BUF_DEP_NEXT, RS_BUF_VEH_COOP, RS_BUF_VEH_COOPLEAD, RS_BUF_ARRIVALS, RS_BUF_DEPARTURES, RS_BUF_MPLIGHT_FULL,
                                RS_BUF_LANE_ARRIVALS, RS_BUF_VEH_COOP_ODD, RS_BUF_VEH_COOPLEAD_ODD, RS_BUF_VEH_MAIL};
extern "C" int rs_snapshot(rs_handle h, void **snap) {
    if (!h || !snap) return RS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, wait_idle(h));
    Snapshot *S = new Snapshot();
    S->seed = h->P.seed;
    for (int b : kSnapBufs) {
        void *d = nullptr;
        if (h->bufs[b].bytes == 0) { S->ptrs.push_back(nullptr); continue; }
        if (hipMalloc(&d, h->bufs[b].bytes) != hipSuccess) { h->err = "rs_snapshot: hipMalloc failed"; rs_snapshot_free(h, S); return RS_ENOMEM; }
        S->ptrs.push_back(d);
        HIPCHK(h, hipMemcpy(d, h->bufs[b].ptr, h->bufs[b].bytes, hipMemcpyDeviceToDevice));
    }
    *snap = S;
    return RS_OK;
}
extern "C" int rs_restore(rs_handle h, const void *snap) {
    if (!h || !snap) return RS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, wait_idle(h));
    const Snapshot *S = (const Snapshot *)snap;
    size_t i = 0;
    for (int b : kSnapBufs) { if (h->bufs[b].bytes) HIPCHK(h, hipMemcpy(h->bufs[b].ptr, S->ptrs[i], h->bufs[b].bytes, hipMemcpyDeviceToDevice)); ++i; }
    h->P.seed = S->seed;        // the vehicles on the network keep the speed factors they were inserted with
    return RS_OK;
}
extern "C" void rs_snapshot_free(rs_handle h, void *snap) {
    if (!snap) return;
    Snapshot *S = (Snapshot *)snap;
    for (void *p : S->ptrs) (void)hipFree(p);
    delete S;
}

extern "C" int rs_timing(rs_handle h, int32_t enable) {
    if (!h) return RS_EINVAL;
    h->timing = enable != 0;
    return RS_OK;
}
extern "C" int rs_timing_read(rs_handle h, float *total_ms, int32_t *launches) {
    if (!h) return RS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    float tot = 0.0f;
    for (size_t i = 0; i < h->ev_used; ++i) {
        float ms = 0.0f;
        HIPCHK(h, hipEventElapsedTime(&ms, h->events[i].first, h->events[i].second));
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = (int32_t)h->ev_used;
    h->ev_used = 0;
    return RS_OK;
}

extern "C" int rs_phase_profile(rs_handle h, int32_t enable, uint64_t *host_out16) {
    if (!h) return RS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    if (h->prof && host_out16) HIPCHK(h, hipMemcpy(host_out16, h->prof, 16 * 8, hipMemcpyDeviceToHost));
    if (enable && !h->prof) { int rc = dev_alloc(h, &h->prof, 16); if (rc) return rc; }
    if (h->prof) HIPCHK(h, hipMemset(h->prof, 0, 16 * 8));
    if (!enable) h->prof = nullptr;     // the allocation stays on the handle's free list
    return RS_OK;
}

extern "C" int rs_set_seed(rs_handle h, uint32_t seed) {
    if (!h) return RS_EINVAL;
    h->P.seed = seed;
    return RS_OK;
}

extern "C" int rs_info(rs_handle h, int32_t *n_envs, int32_t *block_threads, int32_t *lds_bytes, int32_t *max_lanes_per_signal) {
    if (!h) return RS_EINVAL;
    if (n_envs) *n_envs = h->n_envs;
    if (block_threads) *block_threads = h->block;
    if (lds_bytes) *lds_bytes = (int32_t)h->lds;
    if (max_lanes_per_signal) *max_lanes_per_signal = h->K.lmax;
    return RS_OK;
}


// ------------------------------------------------------------------------------------------------ fused IDQN policy
struct rs_policy {
    int device = 0;
    PolicyTab W{};
    std::vector<void *> allocs;
};

template <class T> static int pol_upload(rs_policy *p, const T **dst, const void *src, size_t count) {
    void *d = nullptr;
    if (hipMalloc(&d, count * sizeof(T) + 2048) != hipSuccess) return RS_ENOMEM;     // the fc1 copy passes may read up to 1 KB past the end
    p->allocs.push_back(d);
    if (hipMemcpy(d, src, count * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return RS_EHIP;
    *dst = (const T *)d;
    return RS_OK;
}

extern "C" int rs_idqn_create(int32_t device_id, int32_t n_signals, int32_t lmax, const int32_t *n_actions, const float *conv_w,
                              const float *conv_b, const uint16_t *w1, const float *b1, const uint16_t *w2, const float *b2,
                              const uint16_t *w3, const float *b3, rs_policy_handle *out) {
    if (!out) return RS_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_create_err = "no HIP device visible (this library has no CPU fallback)"; return RS_EHIP; }
    if (device_id < 0 || device_id >= ndev || n_signals <= 0 || lmax < 2 || lmax > 17 || !n_actions || !conv_w || !conv_b || !w1 || !b1 || !w2 || !b2 || !w3 || !b3) {
        g_create_err = "rs_idqn_create: bad argument (1 <= signals, 2 <= lmax <= 17)"; return RS_EINVAL;
    }
    for (int s = 0; s < n_signals; ++s)
        if (n_actions[s] < 1 || n_actions[s] > POL_QMAX) { g_create_err = "rs_idqn_create: 1..8 actions per signal"; return RS_ELIMIT; }
    if (hipSetDevice(device_id) != hipSuccess) { g_create_err = "hipSetDevice failed"; return RS_EHIP; }
    rs_policy *p = new (std::nothrow) rs_policy();
    if (!p) return RS_ENOMEM;
    p->device = device_id;
    const size_t S = (size_t)n_signals, hp = (size_t)(lmax / 2);       // ceil((lmax - 1) / 2)
    p->W.S = n_signals; p->W.lmax = lmax; p->W.hp = (int32_t)hp;
    int rc;
    if ((rc = pol_upload<float>(p, &p->W.conv_w, conv_w, S * 64 * 4)) || (rc = pol_upload<float>(p, &p->W.conv_b, conv_b, S * 64)) ||
        (rc = pol_upload<h4_t>(p, &p->W.w1, w1, S * 64 * hp * 2 * 64)) || (rc = pol_upload<float>(p, &p->W.b1, b1, S * 64)) ||
        (rc = pol_upload<h4_t>(p, &p->W.w2, w2, S * 8 * 2 * 64)) || (rc = pol_upload<float>(p, &p->W.b2, b2, S * 64)) ||
        (rc = pol_upload<h4_t>(p, &p->W.w3, w3, S * 8 * 64)) || (rc = pol_upload<float>(p, &p->W.b3, b3, S * 32)) ||
        (rc = pol_upload<int32_t>(p, &p->W.n_actions, n_actions, S)) ||
        (rc = pol_upload<int32_t>(p, &p->W.hp_sig, std::vector<int32_t>(S, (int32_t)hp).data(), S))) {    // every k-step until rs_idqn_set_lanes says otherwise
        g_create_err = "rs_idqn_create: device allocation / upload failed";
        rs_idqn_destroy(p);
        return rc;
    }
    *out = p;
    return RS_OK;
}

extern "C" int rs_idqn_act(rs_policy_handle p, const void *obs, int32_t n_envs, int32_t env_base, int32_t mode, float epsilon, uint32_t seed, uint32_t step_key,
                           const void *dyn, int32_t *actions, float *q, void *stream) {
    if (!p || !obs || !actions || n_envs <= 0 || mode < 0 || mode > 1) return RS_EINVAL;
    if (hipSetDevice(p->device) != hipSuccess) return RS_EHIP;
    hipLaunchKernelGGL(rs_idqn_forward_kernel, dim3((n_envs + POL_TM - 1) / POL_TM, p->W.S), dim3(256), 0, (hipStream_t)stream,
                       p->W, (const __half *)obs, (int)n_envs, (int)env_base, (int)mode, epsilon, seed, step_key, (const uint32_t *)dyn, actions, q);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_EHIP;
}

// ---- one env-step (or n of them) of a whole group of handles in ONE call (include/resco_sim.h: rs_group_step)
extern "C" int rs_group_step(const rs_handle *hs, int32_t n_handles, const rs_group_agent *agent, int32_t n_steps) {
    if (!hs || n_handles <= 0 || n_steps <= 0) return RS_EINVAL;
    const int kind = agent ? agent->kind : RS_AGENT_NONE;
    for (int i = 0; i < n_handles; ++i) {
        rs_sim *h = hs[i];
        if (!h) return RS_EINVAL;
        if (kind == RS_AGENT_MAXWAVE || kind == RS_AGENT_MAXPRESSURE) {
            if (!h->pairs) { h->err = "rs_group_step: the MAXWAVE / MAXPRESSURE tables are installed by a first rs_act_maxwave call"; return RS_EINVAL; }
            if (!(h->out_mask & (kind == RS_AGENT_MAXPRESSURE ? OUT_MPLIGHT : OUT_WAVE))) { h->err = "rs_group_step: the agent's input buffer is switched off (rs_set_outputs)"; return RS_EINVAL; }
        } else if (kind == RS_AGENT_IDQN) {
            if (!agent->policy || agent->mode < 0 || agent->mode > 1 || agent->policy->W.S != h->K.n_signals || agent->policy->W.lmax != h->K.lmax) {
                h->err = "rs_group_step: RS_AGENT_IDQN needs a policy built for this scenario (n_signals, lmax)"; return RS_EINVAL; }
            if (agent->policy->device != h->device) { h->err = "rs_group_step: the policy's weights live on another device than this handle"; return RS_EINVAL; }
            if (!(h->out_mask & OUT_DRQ_F16)) { h->err = "rs_group_step: RS_AGENT_IDQN reads RS_BUF_DRQ_NORM_F16, which rs_set_outputs has switched off"; return RS_EINVAL; }
        } else if (kind != RS_AGENT_NONE && kind != RS_AGENT_RANDOM) { h->err = "rs_group_step: unknown agent kind"; return RS_EINVAL; }
    }
    for (int k = 0; k < n_steps; ++k)
        for (int i = 0; i < n_handles; ++i) {
            rs_sim *h = hs[i];
            HIPCHK(h, hipSetDevice(h->device));
            hipStream_t st = h->stream;
            h->last = st;
            const int total = h->n_envs * h->K.n_signals;
            if (kind == RS_AGENT_RANDOM)
                hipLaunchKernelGGL(rs_act_random_kernel, dim3((total + 255) / 256), dim3(256), 0, st, h->K, h->P, agent->step_key + (uint32_t)k, h->actions);
            else if (kind == RS_AGENT_MAXWAVE || kind == RS_AGENT_MAXPRESSURE)
                hipLaunchKernelGGL(rs_act_maxwave_kernel, dim3((total + 255) / 256), dim3(256), 0, st, h->K, h->P, (const int32_t *)h->pairs,
                                   h->n_pairs, (const int32_t *)h->valid, (const int32_t *)h->order, (int)(kind == RS_AGENT_MAXPRESSURE),
                                   (const int32_t *)h->O.mplight(), (const int32_t *)h->O.wave(), h->actions);
            else if (kind == RS_AGENT_IDQN) {
                float eps = agent->epsilon + (float)k * agent->epsilon_step;
                if (eps < 0.0f) eps = 0.0f;
                hipLaunchKernelGGL(rs_idqn_forward_kernel, dim3((h->n_envs + POL_TM - 1) / POL_TM, h->K.n_signals), dim3(256), 0, st,
                                   agent->policy->W, (const __half *)h->O.drq_f16(), (int)h->n_envs, (int)h->P.env_base, (int)agent->mode, eps,
                                   agent->seed, agent->step_key + (uint32_t)k, (const uint32_t *)nullptr, h->actions, (float *)nullptr);
            }
            if (kind != RS_AGENT_NONE) HIPCHK(h, hipGetLastError());
            const int rc = launch_step(h, st, h->K.step_length * h->ratio, 1);
            if (rc != RS_OK) return rc;
        }
    return RS_OK;
}

extern "C" int rs_idqn_set_device_weights(rs_policy_handle p, const float *conv_w, const float *conv_b, const uint16_t *w1, const float *b1,
                                          const uint16_t *w2, const float *b2, const uint16_t *w3, const float *b3) {
    if (!p) return RS_EINVAL;
    if (conv_w) p->W.conv_w = conv_w;
    if (conv_b) p->W.conv_b = conv_b;
    if (w1) p->W.w1 = (const h4_t *)w1;
    if (b1) p->W.b1 = b1;
    if (w2) p->W.w2 = (const h4_t *)w2;
    if (b2) p->W.b2 = b2;
    if (w3) p->W.w3 = (const h4_t *)w3;
    if (b3) p->W.b3 = b3;
    return RS_OK;
}

// The networks' own input sizes: signal s observes lanes[s] lanes, so the fc1 rows of its padded lanes (beyond (lanes[s] - 1) * 4
// per conv channel) are zero and the kernel may skip their k-steps -- results are unchanged, the work follows the real head sizes.
extern "C" int rs_idqn_set_lanes(rs_policy_handle p, const int32_t *lanes_per_signal) {
    if (!p || !lanes_per_signal) return RS_EINVAL;
    if (hipSetDevice(p->device) != hipSuccess) return RS_EHIP;
    std::vector<int32_t> hp((size_t)p->W.S);
    for (int s = 0; s < p->W.S; ++s) {
        const int l = lanes_per_signal[s];
        if (l < 2 || l > p->W.lmax) { g_create_err = "rs_idqn_set_lanes: 2 <= lanes[s] <= lmax"; return RS_EINVAL; }
        hp[(size_t)s] = l / 2;                                  // ceil((l - 1) / 2)
    }
    if (hipDeviceSynchronize() != hipSuccess) return RS_EHIP;
    return hipMemcpy((void *)p->W.hp_sig, hp.data(), hp.size() * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess ? RS_OK : RS_EHIP;
}

extern "C" void rs_idqn_destroy(rs_policy_handle p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();
    for (void *d : p->allocs) (void)hipFree(d);
    delete p;
}
