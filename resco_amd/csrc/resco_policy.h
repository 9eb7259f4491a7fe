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
// One workgroup = 2 waves = 64 environments of ONE signal (grid: ceil(N/64) x S); a wave owns 32 rows (envs) x 64
// fc1 outputs = two 32x32 accumulator tiles.  fc2 / fc3 reuse the same MFMA shape after an LDS round trip that
// turns accumulator layout into A-fragment layout; the epilogue does the masked argmax and the epsilon-greedy draw
// (counter hash over (seed; env_base + env, signal, step): rs_group_step passes the handle's first global environment index, so
// that a batch split over pipes draws what the single batch draws) - or, in mode 1, samples from softmax(outputs), which is the IPPO
// policy head on the same trunk - and writes int32 actions the step kernel consumes.
//
// Round 5 measured the 64-channel loop split over two wave pairs (4 waves, channels 0-31 / 32-63, partial sums through LDS): 62 KB
// of LDS instead of 46 KB per workgroup.  The policy of one pipe runs under the step kernels of the others, whose workgroups hold
// 53 KB each, three to a CU: a 46 KB workgroup starts as soon as ONE of them retires, a 62 KB one needs two.  A/B on one box,
// ingolstadt21, sim + policy / sim-only (tools/pipes_ab.py --group, profiles/r05_pipes_group.txt): 1024 envs x 4 pipes 0.76 instead
// of 0.83, x 8 pipes 0.74 instead of 0.85, 4096 envs x 2 pipes 0.86 instead of 0.91.  Not adopted.
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

struct PolicySmem {
    __attribute__((aligned(16))) _Float16 xs[POL_TM][18][8];       // obs tile, rows padded to 8 halfs, +1 zero row
    __attribute__((aligned(16))) _Float16 ys[2][32][POL_C + 8];    // per wave: activations for the next layer's A fragments
    float qs[2][32][POL_QMAX];
    __attribute__((aligned(16))) h4_t wbuf[2 * 8 * 2 * 64];        // fc1 fragments of two conv channels
};

// HP = the k-steps of 8 per conv channel THIS signal needs (ceil((lanes_s - 1) / 2)) as a template parameter: the k-steps become
// straight-line code, so the LDS fragment reads of later steps are issued ahead of the MFMAs of earlier ones.  The packed fc1
// fragments keep the layout of the widest signal (W.hp k-steps per channel); a narrower signal reads the first HP of them -- the
// others belong to padded lanes, whose fc1 rows are zero: skipping them changes no result.
template <int HP>
__device__ __forceinline__ void
idqn_forward_body(PolicySmem &sm, const PolicyTab &W, const __half *__restrict__ obs, int n_envs, int env_base, int mode, float eps, uint32_t seed,
                  uint32_t step_key, int32_t *__restrict__ actions, float *__restrict__ q_out) {
    auto &xs = sm.xs; auto &ys = sm.ys; auto &qs = sm.qs; auto &wbuf = sm.wbuf;
    const int s = blockIdx.y;
    const int m0 = blockIdx.x * POL_TM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, g = lane >> 5;
    const int LM = W.lmax, H = LM - 1;

    // ---- stage the observation tile: [64 envs][LM][5] halfs, contiguous per env -> xs[env][row][0..7] (five values + zeros).
    //      One thread per (env, row): five 2-byte loads, ONE 16-byte LDS store that also clears the row's padding.  Only the rows
    //      this signal's k-steps read (0 .. 2 HP) are staged; rows the tensor does not have, and environments beyond the batch,
    //      are zero.
    constexpr int NR = 2 * HP + 1;
    const int nrow = LM < NR ? LM : NR;
    for (int p = tid; p < POL_TM * NR; p += 128) {
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
    const int mrow = wave * 32 + i;
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

    // ---- conv + ReLU on the fly -> fc1 (two 32x32 tiles per wave).  The fc1 fragments of one conv channel (HP x 2 x 64
    //      lanes x 8 B = 8 KB for H = 16) are shared by the two waves: staged through LDS, double-buffered, the next
    //      channel's global loads in flight while this one is multiplied.
    f16x_t acc0 = {0}, acc1 = {0};
    const float *cw = W.conv_w + (size_t)s * POL_C * 4;
    const float *cb = W.conv_b + (size_t)s * POL_C;
    const int chunk16 = W.hp * 64;                                  // 16-byte units per channel in memory: W.hp * 2 * 64 * 8 B / 16
    constexpr int NQ = (HP * 64 + 127) / 128;                       // copy passes of the 128 threads (the last one may run
                                                                    // past the channel: the allocation and wbuf are padded)
    const uint4 *w1g = (const uint4 *)(W.w1 + (size_t)s * POL_C * W.hp * 2 * 64);
    uint4 *wbuf16 = (uint4 *)wbuf;
#pragma unroll
    for (int q = 0; q < NQ; ++q) wbuf16[tid + q * 128] = w1g[tid + q * 128];
    __syncthreads();
    for (int c = 0; c < POL_C; ++c) {
        uint4 nxt[NQ];
        const uint4 *gn = w1g + (size_t)(c + 1 < POL_C ? c + 1 : c) * chunk16 + tid;
#pragma unroll
        for (int q = 0; q < NQ; ++q) nxt[q] = gn[q * 128];          // in flight while this channel is multiplied
        const _Float16 s00 = (_Float16)cw[c * 4 + 0], s01 = (_Float16)cw[c * 4 + 1], s10 = (_Float16)cw[c * 4 + 2], s11 = (_Float16)cw[c * 4 + 3], sb = (_Float16)cb[c];
        const h2_t h00 = {s00, s00}, h01 = {s01, s01}, h10 = {s10, s10}, h11 = {s11, s11}, hb = {sb, sb}, hz = {(_Float16)0.0f, (_Float16)0.0f};
        const h4_t *wc = wbuf + (size_t)(c & 1) * (8 * 2 * 64) + lane;
#pragma unroll
        for (int kk = 0; kk < HP; ++kk) {
            // features (w = 0,1) and (w = 2,3) of row h: rows beyond H give relu(bias), harmless (zero fc1 weights)
            h2_t f01 = __builtin_elementwise_fma(h00, p0[kk][0], hb), f23 = __builtin_elementwise_fma(h00, p0[kk][2], hb);
            f01 = __builtin_elementwise_fma(h01, p0[kk][1], f01); f23 = __builtin_elementwise_fma(h01, p0[kk][3], f23);
            f01 = __builtin_elementwise_fma(h10, p1[kk][0], f01); f23 = __builtin_elementwise_fma(h10, p1[kk][2], f23);
            f01 = __builtin_elementwise_fma(h11, p1[kk][1], f01); f23 = __builtin_elementwise_fma(h11, p1[kk][3], f23);
            f01 = __builtin_elementwise_max(f01, hz); f23 = __builtin_elementwise_max(f23, hz);
            const h4_t a = h4_t{f01[0], f01[1], f23[0], f23[1]};
            const h4_t b0 = wc[(kk * 2 + 0) * 64], b1 = wc[(kk * 2 + 1) * 64];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x8f16(a, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x8f16(a, b1, acc1, 0, 0, 0);
        }
        uint4 *dst = wbuf16 + (size_t)((c + 1) & 1) * (8 * 2 * 64 / 2) + tid;
#pragma unroll
        for (int q = 0; q < NQ; ++q) dst[q * 128] = nxt[q];
        __syncthreads();
    }

    // ---- bias + ReLU, accumulator layout -> LDS [row][col] -> A fragments of the next layer
    _Float16 (*y)[POL_C + 8] = ys[wave];
    {
        const float *b1v = W.b1 + (size_t)s * POL_C;
        const float ba = b1v[i], bb = b1v[32 + i];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * g;
            float u = acc0[r] + ba, v = acc1[r] + bb;
            y[row][i] = (_Float16)(u > 0.0f ? u : 0.0f);
            y[row][32 + i] = (_Float16)(v > 0.0f ? v : 0.0f);
        }
    }
    __syncthreads();
    // ---- fc2
    f16x_t c0 = {0}, c1 = {0};
    {
        const h4_t *w2 = W.w2 + (size_t)s * 8 * 2 * 64 + lane;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const h4_t a = *(const h4_t *)&y[i][kk * 8 + g * 4];
            c0 = __builtin_amdgcn_mfma_f32_32x32x8f16(a, w2[(kk * 2 + 0) * 64], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x8f16(a, w2[(kk * 2 + 1) * 64], c1, 0, 0, 0);
        }
    }
    __syncthreads();
    {
        const float *b2v = W.b2 + (size_t)s * POL_C;
        const float ba = b2v[i], bb = b2v[32 + i];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * g;
            float u = c0[r] + ba, v = c1[r] + bb;
            y[row][i] = (_Float16)(u > 0.0f ? u : 0.0f);
            y[row][32 + i] = (_Float16)(v > 0.0f ? v : 0.0f);
        }
    }
    __syncthreads();
    // ---- fc3 (columns >= n_actions are zero padding)
    f16x_t d0 = {0};
    {
        const h4_t *w3 = W.w3 + (size_t)s * 8 * 64 + lane;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const h4_t a = *(const h4_t *)&y[i][kk * 8 + g * 4];
            d0 = __builtin_amdgcn_mfma_f32_32x32x8f16(a, w3[kk * 64], d0, 0, 0, 0);
        }
    }
    if (i < POL_QMAX) {
        const float b3v = W.b3[(size_t)s * 32 + i];
#pragma unroll
        for (int r = 0; r < 16; ++r) qs[wave][(r & 3) + 8 * (r >> 2) + 4 * g][i] = d0[r] + b3v;
    }
    __syncthreads();
    // ---- per environment: greedy action over the signal's actions, epsilon-greedy draw
    if (lane < 32) {
        const int m = m0 + wave * 32 + lane;
        if (m < n_envs) {
            const int na = W.n_actions[s];
            int best = 0;
            float bq = qs[wave][lane][0];
            for (int a = 1; a < na; ++a) { const float v = qs[wave][lane][a]; if (v > bq) { bq = v; best = a; } }
            int act = best;
            if (mode == 1) {
                // categorical policy (the IPPO head, pfrl_ppo.py:57-60): the outputs are logits, a ~ softmax(logits)
                float z = 0.0f;
                for (int a = 0; a < na; ++a) z += __expf(qs[wave][lane][a] - bq);
                const float u = d_u01(pol_hash(seed ^ 0x1D0A17u, (uint32_t)(env_base + m), (uint32_t)s, step_key, 2u)) * z;
                float cum = 0.0f;
                act = na - 1;
                for (int a = 0; a < na; ++a) { cum += __expf(qs[wave][lane][a] - bq); if (u < cum) { act = a; break; } }
            } else if (eps > 0.0f) {
                const float u = d_u01(pol_hash(seed ^ 0x1D0A17u, (uint32_t)(env_base + m), (uint32_t)s, step_key, 0u));
                if (u < eps) act = (int)(pol_hash(seed ^ 0x1D0A17u, (uint32_t)(env_base + m), (uint32_t)s, step_key, 1u) % (uint32_t)na);
            }
            actions[(size_t)m * W.S + s] = act;
            if (q_out)
                for (int a = 0; a < POL_QMAX; ++a) q_out[((size_t)m * W.S + s) * POL_QMAX + a] = a < na ? qs[wave][lane][a] : -INFINITY;
        }
    }
}

// One workgroup = one signal: the body is chosen by the signal's own head size (a workgroup-uniform switch).  W.hp_sig[s] == W.hp
// for every signal until rs_idqn_set_lanes has told the library the networks' real input sizes.
__global__ void __launch_bounds__(128)
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
