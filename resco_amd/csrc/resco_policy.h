// resco_policy.h -- fused forward of the S per-signal IDQN networks (BASELINE config 5).
//
// The reference network per signal (resco_benchmark/agents/pfrl_dqn.py:30-39):
//     Conv2d(1, 64, (2,2)) - ReLU - Flatten(c,h,w) - Linear(64*H*4, 64) - ReLU - Linear(64, 64) - ReLU - Linear(64, A)
// on the observation (1, L, 5), H = L - 1.  As separate library calls the 64*H*4-wide feature tensor (4 096 halfs
// per env and signal) goes out to HBM and comes back - that traffic, not the 11 GFLOP of fc1, is what the PyTorch
// forward spends its time on.  Here the features never exist in memory: for every conv channel a lane computes the
// 4 features its MFMA A-fragment needs from the 2 x 5 observation values it keeps in registers, and feeds them
// straight into v_mfma_f32_32x32x8_f16 against the pre-packed fc1 weights of that channel.
//
// One workgroup = 4 waves = 64 environments of ONE signal (grid: ceil(N/64) x S); wave (mh, nh) owns rows (envs) 32 mh .. 32 mh + 31
// and fc1 outputs 32 nh .. 32 nh + 31 = ONE 32x32 accumulator tile (the two waves of a row half compute the same conv features: ten
// packed FMAs per k-step next to a 64-cycle MFMA).  fc2 / fc3 reuse the same MFMA shape after an LDS round trip that
// turns accumulator layout into A-fragment layout; the epilogue does the masked argmax and the epsilon-greedy draw
// (counter hash over (seed; env_base + env, signal, step): rs_group_step passes the handle's first global environment index, so
// that a batch split over pipes draws what the single batch draws) - or, in mode 1, samples from softmax(outputs), which is the IPPO
// policy head on the same trunk - and writes int32 actions the step kernel consumes.
//
// Round 5, the shape of the workgroup (one MI355X, ingolstadt21, kernel trace of the launch for 1024 / 4096 environments).  Rounds
// 1-4 ran 2 waves, each with BOTH fc1 tiles of its 32 rows, the fc1 fragments of a conv channel staged through LDS for the workgroup
// one channel ahead, a barrier per channel: 53.4 / 88.5 us, 46 KB of LDS.  That was 64 dependent round trips to L2, not MFMA time:
//   * the 64-channel loop split over two wave pairs (channels 0-31 / 32-63, partial sums through LDS; 62 KB): the policy of one pipe
//     runs under the step kernels of the others, whose workgroups hold 53 KB each, three to a CU -- a 46 KB workgroup starts as soon
//     as ONE of them retires, a 62 KB one needs two: sim + policy / sim-only 0.76 instead of 0.83 at 1024 envs x 4 pipes.  Dropped.
//   * the two fc1 TILES of a row half on two waves, staging unchanged: 53 us again -- the loop waits for the loads, not the MFMAs.
//   * this file: no staging buffer and no barrier in the loop; every wave reads the fragments of its tile from global memory into a
//     ring of registers ~16 k-steps ahead, the conv weights come from LDS, two accumulators alternate: 30.8 / 68.7 us, 21 KB of LDS
//     (the observation tile and the activations share their memory), 196 VGPRs.  A ring of 32 k-steps is 2 us faster alone and
//     needs 236 VGPRs -- more than the 224 a SIMD has left beside the step waves of two resident workgroups: 0.75 instead of 0.87.
//   sim + policy / sim-only through rs_group_step: 1024 envs x 4 pipes 0.833 -> 0.867, x 8 pipes 0.853 -> 0.873, 4096 x 2 0.91 -> 0.93.
//
// MFMA 32x32x8 f16 fragment layout (lane l, g = l >> 5, i = l & 31):
//     A: row i, k = 4 g + j (j = 0..3);   B: column i, k = 4 g + j;   C/D reg r: column i, row (r & 3) + 8 (r >> 2) + 4 g.
#pragma once

typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef float f16x_t __attribute__((ext_vector_type(16)));

#define POL_C 64            // conv channels = fc widths of the reference network
#define POL_TM 64           // environments per workgroup
#define POL_QMAX 8          // at most 8 actions per signal
#ifndef POL_SB
#define POL_SB 0x387        // scheduling barrier of the fc1 loop: ALU, VALU, SALU and DS instructions may cross it; VMEM and MFMA may not
#endif
#ifndef POL_RING
#define POL_RING 16         // fc1 B fragments (k-steps) in flight ahead of the one being multiplied
#endif

struct PolicyTab {          // device pointers of the packed weights, all [S][...]
    const float *conv_w;    // [S][64][4]  w00 w01 w10 w11
    const float *conv_b;    // [S][64]
    const h4_t *w1;         // [S][64 channels][HP k-steps][2 n-tiles][64 lanes]  B fragments of fc1
    const float *b1;        // [S][64]
    const h4_t *w2;         // [S][8][2][64]
    const float *b2;        // [S][64]
    const h4_t *w3;         // [S][8][64]       (columns >= n_actions are zero)
    const float *b3;        // [S][32]
    const int32_t *n_actions;   // [S]
    const int32_t *hp_sig;      // [S] k-steps of 8 that carry non-zero fc1 rows for signal s = ceil((lanes_s - 1) / 2), <= hp (rs_idqn_set_lanes)
    int32_t S, lmax, hp;    // signals, padded lanes per signal (obs rows), k-steps of 8 per channel = ceil((lmax-1)/2)
};

__device__ __forceinline__ uint32_t pol_hash(uint32_t seed, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    return d_hash(seed, a, b, c, d);
}

struct PolicySmem {         // 18 KB: the observation tile is dead once the lanes hold their rows, the activations live in its place
    union {
        __attribute__((aligned(16))) _Float16 xs[POL_TM][18][8];       // obs tile, rows padded to 8 halfs, +1 zero row
        struct {
            __attribute__((aligned(16))) _Float16 ys[2][32][POL_C + 8];    // per row half: activations for the next layer's A fragments
            float qs[2][32][POL_QMAX];
            h2_t cwp[POL_C][8];         // the signal's conv weights as packed pairs: w00 w00 | w01 w01 | w10 w10 | w11 w11 | b b | - - -
        };
    };
};

// HP = the k-steps of 8 per conv channel THIS signal needs (ceil((lanes_s - 1) / 2)) as a template parameter: the k-steps become
// straight-line code, so the LDS fragment reads of later steps are issued ahead of the MFMAs of earlier ones.  The packed fc1
// fragments keep the layout of the widest signal (W.hp k-steps per channel); a narrower signal reads the first HP of them -- the
// others belong to padded lanes, whose fc1 rows are zero: skipping them changes no result.
template <int HP>
__device__ __forceinline__ void
idqn_forward_body(PolicySmem &sm, const PolicyTab &W, const __half *__restrict__ obs, int n_envs, int env_base, int mode, float eps, uint32_t seed,
                  uint32_t step_key, int32_t *__restrict__ actions, float *__restrict__ q_out) {
    auto &xs = sm.xs; auto &ys = sm.ys; auto &qs = sm.qs; auto &cwp = sm.cwp;
    const int s = blockIdx.y;
    const int m0 = blockIdx.x * POL_TM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mh = wave & 1, nh = wave >> 1;                        // row half, fc tile of this wave
    const int i = lane & 31, g = lane >> 5;
    const int LM = W.lmax, H = LM - 1;

    // ---- stage the observation tile: [64 envs][LM][5] halfs, contiguous per env -> xs[env][row][0..7] (five values + zeros).
    //      One thread per (env, row): five 2-byte loads, ONE 16-byte LDS store that also clears the row's padding.  Only the rows
    //      this signal's k-steps read (0 .. 2 HP) are staged; rows the tensor does not have, and environments beyond the batch,
    //      are zero.
    constexpr int NR = 2 * HP + 1;
    const int nrow = LM < NR ? LM : NR;
    for (int p = tid; p < POL_TM * NR; p += 256) {
        const int m = p / NR, row = p - m * NR;
        union { uint4 q; _Float16 h[8]; } u;
        u.q = uint4{0u, 0u, 0u, 0u};
        if (row < nrow && m0 + m < n_envs) {
            const _Float16 *src = (const _Float16 *)obs + ((size_t)(m0 + m) * W.S + s) * LM * 5 + row * 5;
#pragma unroll
            for (int j = 0; j < 5; ++j) u.h[j] = src[j];
        }
        *(uint4 *)&xs[m][row][0] = u.q;
    }
    __syncthreads();

    // ---- my rows of the tile in registers: for k-step kk the lane needs obs rows h = 2 kk + g and h + 1, as the four
    //      overlapping column pairs (0,1) (1,2) (2,3) (3,4) of each row: the 2x2 convolution then is 8 packed f16 FMAs
    //      per k-step (v_pk_fma_f16: two features per instruction) and its result already is the A fragment
    const int mrow = mh * 32 + i;
    h2_t p0[HP][4], p1[HP][4];
#pragma unroll
    for (int kk = 0; kk < HP; ++kk) {
        const int h = 2 * kk + g;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const _Float16 a0 = h < H ? xs[mrow][h][j] : (_Float16)0.0f, a1 = h < H ? xs[mrow][h][j + 1] : (_Float16)0.0f;
            const _Float16 b0 = h < H ? xs[mrow][h + 1][j] : (_Float16)0.0f, b1 = h < H ? xs[mrow][h + 1][j + 1] : (_Float16)0.0f;
            p0[kk][j] = h2_t{a0, a1};
            p1[kk][j] = h2_t{b0, b1};
        }
    }

    __syncthreads();                            // (xs is dead from here on: ys / qs live in the same memory)

    // ---- conv + ReLU on the fly -> fc1 (one 32x32 tile per wave).  The wave reads the B fragments of ITS tile straight from
    //      global memory (HP coalesced 512-byte loads per conv channel; the other row half's wave reads the same lines) into a ring
    //      of registers, D channels = ~POL_RING k-steps ahead of the one being multiplied: a k-step is one MFMA of 64 cycles, a load
    //      from L2 takes 10-20 of those.  (Rounds 1-4 staged a channel's fragments through LDS for the whole workgroup, ONE channel
    //      ahead, with a barrier per channel: 53 us per launch whatever the wave layout, the latency of 64 dependent round trips.)
    //      The order -- multiply channel c out of ring slot d, THEN refill slot d with channel c + D -- is pinned with scheduling
    //      barriers: left alone the scheduler hoists the refills and ends every iteration on s_waitcnt vmcnt(0).
    f16x_t acc0 = {0}, acc1 = {0};
    if (tid < POL_C) {          // (in LDS: a scalar load per channel could not be moved across the scheduling barriers below)
        const float *cw = W.conv_w + ((size_t)s * POL_C + tid) * 4;
        const _Float16 b = (_Float16)W.conv_b[(size_t)s * POL_C + tid];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const _Float16 w = (_Float16)cw[j]; cwp[tid][j] = h2_t{w, w}; }
        cwp[tid][4] = h2_t{b, b};
    }
    __syncthreads();
    const h4_t *wg = W.w1 + ((size_t)s * POL_C * W.hp * 2 + nh) * 64 + lane;       // fragment (c, kk) of this tile: + (c * W.hp + kk) * 128
    const int cstride = W.hp * 128;
    constexpr int D = HP >= 5 ? (POL_RING >= 32 ? 4 : 2) : (HP >= 3 ? POL_RING / 4 : (HP == 2 ? POL_RING / 2 : POL_RING));
    static_assert(POL_C % D == 0 && D * HP <= 64, "ring size");
    h4_t bq[D][HP];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int kk = 0; kk < HP; ++kk) bq[d][kk] = wg[(size_t)d * cstride + kk * 128];
    for (int c0 = 0; c0 < POL_C; c0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int c = c0 + d;
            const h2_t h00 = cwp[c][0], h01 = cwp[c][1], h10 = cwp[c][2], h11 = cwp[c][3], hb = cwp[c][4], hz = {(_Float16)0.0f, (_Float16)0.0f};
#pragma unroll
            for (int kk = 0; kk < HP; ++kk) {
                // features (w = 0,1) and (w = 2,3) of row h: rows beyond H give relu(bias), harmless (zero fc1 weights)
                h2_t f01 = __builtin_elementwise_fma(h00, p0[kk][0], hb), f23 = __builtin_elementwise_fma(h00, p0[kk][2], hb);
                f01 = __builtin_elementwise_fma(h01, p0[kk][1], f01); f23 = __builtin_elementwise_fma(h01, p0[kk][3], f23);
                f01 = __builtin_elementwise_fma(h10, p1[kk][0], f01); f23 = __builtin_elementwise_fma(h10, p1[kk][2], f23);
                f01 = __builtin_elementwise_fma(h11, p1[kk][1], f01); f23 = __builtin_elementwise_fma(h11, p1[kk][3], f23);
                f01 = __builtin_elementwise_max(f01, hz); f23 = __builtin_elementwise_max(f23, hz);
                const h4_t a = h4_t{f01[0], f01[1], f23[0], f23[1]};
                // (two accumulators, alternating: a chain of MFMAs on ONE accumulator waits for every result -- 34.8 against 30.8 us)
                if ((kk + d * HP) & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x8f16(a, bq[d][kk], acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x8f16(a, bq[d][kk], acc0, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(POL_SB);
            const int cn = c + D < POL_C ? c + D : POL_C - 1;                  // (the last D channels re-read the last one: no branch)
#pragma unroll
            for (int kk = 0; kk < HP; ++kk) bq[d][kk] = wg[(size_t)cn * cstride + kk * 128];
            __builtin_amdgcn_sched_barrier(POL_SB);
        }
    }

#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] += acc1[r];
    // ---- bias + ReLU, accumulator layout -> LDS [row][col] -> A fragments of the next layer
    _Float16 (*y)[POL_C + 8] = ys[mh];           // (the two waves of a row half write their own 32 columns)
    {
        const float ba = W.b1[(size_t)s * POL_C + nh * 32 + i];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * g;
            const float u = acc0[r] + ba;
            y[row][nh * 32 + i] = (_Float16)(u > 0.0f ? u : 0.0f);
        }
    }
    __syncthreads();
    // ---- fc2
    f16x_t c0 = {0};
    {
        const h4_t *w2 = W.w2 + (size_t)s * 8 * 2 * 64 + lane;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const h4_t a = *(const h4_t *)&y[i][kk * 8 + g * 4];
            c0 = __builtin_amdgcn_mfma_f32_32x32x8f16(a, w2[(kk * 2 + nh) * 64], c0, 0, 0, 0);
        }
    }
    __syncthreads();
    {
        const float ba = W.b2[(size_t)s * POL_C + nh * 32 + i];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * g;
            const float u = c0[r] + ba;
            y[row][nh * 32 + i] = (_Float16)(u > 0.0f ? u : 0.0f);
        }
    }
    __syncthreads();
    // ---- fc3 (columns >= n_actions are zero padding): one tile per row half, on its first wave
    if (nh == 0) {
        f16x_t d0 = {0};
        const h4_t *w3 = W.w3 + (size_t)s * 8 * 64 + lane;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const h4_t a = *(const h4_t *)&y[i][kk * 8 + g * 4];
            d0 = __builtin_amdgcn_mfma_f32_32x32x8f16(a, w3[kk * 64], d0, 0, 0, 0);
        }
        if (i < POL_QMAX) {
            const float b3v = W.b3[(size_t)s * 32 + i];
#pragma unroll
            for (int r = 0; r < 16; ++r) qs[mh][(r & 3) + 8 * (r >> 2) + 4 * g][i] = d0[r] + b3v;
        }
    }
    __syncthreads();
    // ---- per environment: greedy action over the signal's actions, epsilon-greedy draw
    if (nh == 0 && lane < 32) {
        const int m = m0 + mh * 32 + lane;
        if (m < n_envs) {
            const int na = W.n_actions[s];
            int best = 0;
            float bq = qs[mh][lane][0];
            for (int a = 1; a < na; ++a) { const float v = qs[mh][lane][a]; if (v > bq) { bq = v; best = a; } }
            int act = best;
            if (mode == 1) {
                // categorical policy (the IPPO head, pfrl_ppo.py:57-60): the outputs are logits, a ~ softmax(logits)
                float z = 0.0f;
                for (int a = 0; a < na; ++a) z += __expf(qs[mh][lane][a] - bq);
                const float u = d_u01(pol_hash(seed ^ 0x1D0A17u, (uint32_t)(env_base + m), (uint32_t)s, step_key, 2u)) * z;
                float cum = 0.0f;
                act = na - 1;
                for (int a = 0; a < na; ++a) { cum += __expf(qs[mh][lane][a] - bq); if (u < cum) { act = a; break; } }
            } else if (eps > 0.0f) {
                const float u = d_u01(pol_hash(seed ^ 0x1D0A17u, (uint32_t)(env_base + m), (uint32_t)s, step_key, 0u));
                if (u < eps) act = (int)(pol_hash(seed ^ 0x1D0A17u, (uint32_t)(env_base + m), (uint32_t)s, step_key, 1u) % (uint32_t)na);
            }
            actions[(size_t)m * W.S + s] = act;
            if (q_out)
                for (int a = 0; a < POL_QMAX; ++a) q_out[((size_t)m * W.S + s) * POL_QMAX + a] = a < na ? qs[mh][lane][a] : -INFINITY;
        }
    }
}

// One workgroup = one signal: the body is chosen by the signal's own head size (a workgroup-uniform switch).  W.hp_sig[s] == W.hp
// for every signal until rs_idqn_set_lanes has told the library the networks' real input sizes.
#ifndef POL_MINBLOCKS
#define POL_MINBLOCKS 1      // workgroups per CU the register allocation leaves room for (study switch: 4 = 128 registers per lane)
#endif
__global__ void __launch_bounds__(256, POL_MINBLOCKS)
rs_idqn_forward_kernel(PolicyTab W, const __half *__restrict__ obs, int n_envs, int env_base, int mode, float eps, uint32_t seed, uint32_t step_key,
                       const uint32_t *__restrict__ dyn, int32_t *__restrict__ actions, float *__restrict__ q_out) {
    // dyn != NULL: epsilon (float bits) and step key come from device memory, so that a captured HIP graph of the
    // env-step can be replayed with values an earlier node of the same graph computed
    if (dyn) { eps = __uint_as_float(dyn[0]); step_key = dyn[1]; }
    __shared__ PolicySmem sm;
    switch (__builtin_amdgcn_readfirstlane(W.hp_sig[blockIdx.y])) {
        case 1: idqn_forward_body<1>(sm, W, obs, n_envs, env_base, mode, eps, seed, step_key, actions, q_out); break;
        case 2: idqn_forward_body<2>(sm, W, obs, n_envs, env_base, mode, eps, seed, step_key, actions, q_out); break;
        case 3: idqn_forward_body<3>(sm, W, obs, n_envs, env_base, mode, eps, seed, step_key, actions, q_out); break;
        case 4: idqn_forward_body<4>(sm, W, obs, n_envs, env_base, mode, eps, seed, step_key, actions, q_out); break;
        case 5: idqn_forward_body<5>(sm, W, obs, n_envs, env_base, mode, eps, seed, step_key, actions, q_out); break;
        case 6: idqn_forward_body<6>(sm, W, obs, n_envs, env_base, mode, eps, seed, step_key, actions, q_out); break;
        case 7: idqn_forward_body<7>(sm, W, obs, n_envs, env_base, mode, eps, seed, step_key, actions, q_out); break;
        default: idqn_forward_body<8>(sm, W, obs, n_envs, env_base, mode, eps, seed, step_key, actions, q_out); break;
    }
}
