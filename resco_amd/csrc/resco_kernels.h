// resco_kernels.h -- device side of the MI355X traffic-signal simulator: table records, LDS layout, the
// fused step kernel (rs_step_kernel), reset and static-agent kernels.  Included by resco_sim.hip (one
// translation unit; gfx950 only).  See resco_sim.hip for the overview and DESIGN.md for the phase structure.
#pragma once

// ------------------------------------------------------------------------------------------------ tables
// X(name, ctype, count)
#define RS_TABLES(X)                                                                                              \
    X(lane_len, float, n_lanes) X(lane_vmax, float, n_lanes) X(lane_edge, int32_t, n_lanes)                       \
    X(lane_left, int32_t, n_lanes) X(lane_right, int32_t, n_lanes) X(lane_link_start, int32_t, n_lanes)           \
    X(lane_link_cnt, int32_t, n_lanes) X(lane_obs, int32_t, n_lanes) X(lane_internal, int32_t, n_lanes)           \
    X(link_to_lane, int32_t, n_links) X(link_dest_lane, int32_t, n_links) X(link_to_edge, int32_t, n_links)       \
    X(link_tls, int32_t, n_links) X(link_tls_pos, int32_t, n_links) X(link_minor, int32_t, n_links)               \
    X(link_cont, int32_t, n_links) X(link_foe_start, int32_t, n_links) X(link_foe_cnt, int32_t, n_links)          \
    X(link_via_len, float, n_links) X(link_via1, int32_t, n_links) X(link_via2, int32_t, n_links)                 \
    X(link_from_lane, int32_t, n_links) X(foe_link, int32_t, n_foes) X(edge_lane0, int32_t, n_edges)              \
    X(edge_nlanes, int32_t, n_edges) X(route_start, int32_t, n_routes + 1) X(route_edge, int32_t, n_route_steps)  \
    X(route_tlsdist, float, n_route_steps) X(route_mask1, uint32_t, n_route_steps)                                \
    X(route_mask2, uint32_t, n_route_steps) X(trip_depart, int32_t, n_trips) X(trip_route, int32_t, n_trips)      \
    X(trip_vtype, int32_t, n_trips) X(trips_cum, int32_t, horizon + 2) X(vtype_params, float, n_vtypes * VT_COLS) \
    X(tls_nphase, int32_t, n_signals) X(tls_ngreen, int32_t, n_signals) X(tls_nlinks, int32_t, n_signals)         \
    X(tls_state_off, int32_t, n_signals) X(tls_dur_off, int32_t, n_signals) X(tls_yel_off, int32_t, n_signals)    \
    X(tls_init_phase, int32_t, n_signals) X(tls_states, int32_t, n_tls_states) X(tls_dur, int32_t, n_tls_dur)     \
    X(tls_yellow, int32_t, n_tls_yellow) X(fix_nphase, int32_t, n_signals) X(fix_state_off, int32_t, n_signals)   \
    X(fix_dur_off, int32_t, n_signals) X(fix_init_phase, int32_t, n_signals) X(fix_init_left, int32_t, n_signals) \
    X(fix_states, int32_t, n_fix_states) X(fix_dur, int32_t, n_fix_dur) X(obs_lane, int32_t, n_obs)               \
    X(sig_obs_start, int32_t, n_signals + 1) X(mv_in_start, int32_t, n_signals * 12 + 1)                          \
    X(mv_in_idx, int32_t, n_mv_in) X(mv_out_start, int32_t, n_signals * 12 + 1) X(mv_out_idx, int32_t, n_mv_out)  \
    X(pr_out_start, int32_t, n_signals + 1) X(pr_out_idx, int32_t, n_pr_out)

struct Tab {
#define X(name, type, count) const type *name;
    RS_TABLES(X)
#undef X
    const int32_t *obs_sig;     // observed lane -> signal index (derived)
    int32_t n_lanes, n_links, n_edges, n_routes, n_trips, n_signals, n_obs, n_vtypes;
    int32_t horizon, capacity, step_length, yellow_length, lmax, n_arr;
};

// Kernel arguments are kept SMALL on purpose: every pointer passed by value costs two SGPRs for the whole
// kernel, and beyond ~100 SGPRs the compiler spills them into VGPR lanes (v_writelane / v_readlane around every
// table access).  The per-slot state and the outputs are therefore ONE allocation each, with field addresses
// computed from (base, N*C) where they are used.
struct State {      // env-major SoA in HBM: field[env][slot]
    char *base;
    size_t nc;          // N * C
    int32_t *trip_log;  // [N][n_trips][4] or NULL
    int32_t *env;       // [N][4] t, next_trip, hw, reserved
    int32_t *tls;       // [N][S][3] phase, left, next_phase
    long long *stats;   // [N][10]
    __host__ __device__ float *pos() const { return (float *)base; }
    __host__ __device__ float *speed() const { return (float *)(base + 4 * nc); }
    __host__ __device__ float *accel() const { return (float *)(base + 8 * nc); }
    __host__ __device__ float *tloss() const { return (float *)(base + 12 * nc); }
    __host__ __device__ float *sf() const { return (float *)(base + 16 * nc); }
    __host__ __device__ uint16_t *lane() const { return (uint16_t *)(base + 20 * nc); }
    __host__ __device__ uint16_t *trip() const { return (uint16_t *)(base + 22 * nc); }
    __host__ __device__ uint16_t *cursor() const { return (uint16_t *)(base + 24 * nc); }
    __host__ __device__ uint16_t *swait() const { return (uint16_t *)(base + 26 * nc); }
    __host__ __device__ uint16_t *rwait() const { return (uint16_t *)(base + 28 * nc); }
    __host__ __device__ uint16_t *depart() const { return (uint16_t *)(base + 30 * nc); }
    __host__ __device__ uint16_t *wtot() const { return (uint16_t *)(base + 32 * nc); }
    __host__ __device__ uint8_t *owner() const { return (uint8_t *)(base + 34 * nc); }
    static size_t bytes(size_t nc_) { return 35 * nc_; }
};

struct Out {        // one allocation; n = N, o = n_obs, s = n_signals, lm = lanes of the largest signal
    char *base;
    int32_t n, o, s, lm;
    __host__ __device__ size_t a5() const { return (size_t)n * o * 5 * 4; }      // one [N][n_obs][5] f32 block
    __host__ __device__ size_t ns() const { return (size_t)n * s * 4; }          // one [N][S] 4-byte block
    __host__ __device__ float *lane_agg() const { return (float *)base; }
    __host__ __device__ float *drq_norm() const { return (float *)(base + a5()); }
    __host__ __device__ float *wait() const { return (float *)(base + 2 * a5()); }
    __host__ __device__ float *wait_norm() const { return (float *)(base + 2 * a5() + ns()); }
    __host__ __device__ int32_t *phase() const { return (int32_t *)(base + 2 * a5() + 2 * ns()); }
    __host__ __device__ int32_t *pressure() const { return (int32_t *)(base + 2 * a5() + 3 * ns()); }
    __host__ __device__ int32_t *queue_sum() const { return (int32_t *)(base + 2 * a5() + 4 * ns()); }
    __host__ __device__ int32_t *queue_max() const { return (int32_t *)(base + 2 * a5() + 5 * ns()); }
    __host__ __device__ int32_t *mplight() const { return (int32_t *)(base + 2 * a5() + 6 * ns()); }
    __host__ __device__ int32_t *wave() const { return (int32_t *)(base + 2 * a5() + 19 * ns()); }
    __host__ __device__ __half *drq_f16() const { return (__half *)(base + 2 * a5() + 31 * ns()); }
    __host__ __device__ size_t bytes() const { return 2 * a5() + 31 * ns() + (size_t)n * s * lm * 5 * 2 + 64; }
};

struct KParams {
    uint32_t seed;
    int32_t env_base;
    float max_distance, sigma;
    int32_t speed_dev, fixed_program;
    int32_t n_ticks;        // ticks to simulate in this launch (0: observe only)
    int32_t do_fsm;         // apply prep_phase / set_phase around the ticks
    int32_t n_envs;
    unsigned long long *prof;   // optional [16] per-phase cycle accumulators (rs_phase_profile), NULL = off
#ifdef RS_DIAG
    int32_t diag;           // diagnostic builds only (tools/diag_phases.sh): bit mask of phase parts to skip
#endif
};
#ifdef RS_COUNT
// event counters of a -DRS_COUNT build (tools/count_events.sh): printed by rs_destroy
__device__ unsigned long long g_count[32];
#define COUNT(i_, n_) atomicAdd(&g_count[i_], (unsigned long long)(n_))
#else
#define COUNT(i_, n_)
#endif
#ifdef RS_BARWAIT
// -DRS_BARWAIT build (tools/count_events.sh): time every wave spends inside the barriers of the step kernel
__device__ unsigned long long g_bar[64];    // [0] barrier wait, [1] wave lifetime (wall_clock64 ticks), [2] barriers, [8 + k] wait at the barrier of source line k (see g_bar_line)
__device__ int g_bar_line[56];
#define SYNC() { constexpr int site_ = __COUNTER__; const unsigned long long a_ = wall_clock64(); __syncthreads(); if ((threadIdx.x & 63) == 0) { const unsigned long long w_ = wall_clock64() - a_; atomicAdd(&g_bar[0], w_); atomicAdd(&g_bar[2], 1ull); atomicAdd(&g_bar[8 + (site_ & 31)], w_); g_bar_line[site_ & 31] = __LINE__; } }
#else
#define SYNC() __syncthreads()
#endif
#ifdef RS_DIAG
#define DIAG_SKIP(bit_) (P.diag & (bit_))
#else
#define DIAG_SKIP(bit_) false
#endif

// ------------------------------------------------------------------------------------------------ device math
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ __forceinline__ uint32_t d_hash(uint32_t seed, uint32_t env, uint32_t trip, uint32_t tick, uint32_t stream) {
    uint32_t h = seed;
    uint32_t w[4] = {env, trip, tick, stream};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t k = w[i];
        k *= 0xcc9e2d51u; k = rotl32(k, 15); k *= 0x1b873593u;
        h ^= k; h = rotl32(h, 13); h = h * 5u + 0xe6546b64u;
    }
    h ^= 16u;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ float d_u01(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

// Krauss (SUMO MSCFModel, Euler update, dt = 1 s) [SUMO-K]
__device__ __forceinline__ float d_brake_gap(float v, float b) {
    int steps = (int)(v / b);
    float fs = (float)steps;
    return fs * v - b * fs * (fs + 1.0f) * 0.5f;
}
__device__ __forceinline__ float d_stop_speed(float gap, float b, float tau) {
    float g = gap - 0.001f;
    if (g < 0.0f) return 0.0f;
    float q = 1.0f + 4.0f * ((2.0f * g / b - tau) + tau * tau);
    float n = floorf(0.5f - (tau + sqrtf(q) * -0.5f));
    float h = 0.5f * n * (n - 1.0f) * b + n * b * tau;
    float r = (g - h) / (n + tau);
    return n * b + r;
}
__device__ __forceinline__ float d_free_speed(float dist, float target, float b) {
    if (dist < target) return target;
    float t2 = b + 2.0f * target;
    float y = ((sqrtf(t2 * t2 + 8.0f * b * dist) - b) * 0.5f - target) / b;
    if (y < 0.0f) y = 0.0f;
    float yf = floorf(y);
    float exact = (yf * yf + yf) * 0.5f * b + yf * target + (y > yf ? target : 0.0f);
    float rest = dist - exact;
    if (rest < 0.0f) rest = 0.0f;
    return rest / (yf + 1.0f) + yf * b + target;
}
__device__ __forceinline__ float d_follow_speed(float gap, float vl, float b, float bl, float tau) {
    float bm = b > bl ? b : bl;
    return d_stop_speed(gap + d_brake_gap(vl, bm), b, tau);
}

// ------------------------------------------------------------------------------------------------ packed tables
// The step kernel reads the scenario through 16-byte records (one global_load_dwordx4 per lane / link /
// route step) built by rs_create from the flat rs_scenario arrays.
struct __attribute__((aligned(16))) LaneRec {
    float len, vmax;
    uint16_t link_start;
    uint8_t link_cnt;
    uint8_t flags;          // bit0 junction-internal; bits 2..7 number of lanes of the edge
    uint16_t cell0;         // first list cell of this lane (cells of CELL_LEN metres, floor(len/CELL_LEN)+1 per lane)
    uint16_t edge_lane0;
};
struct __attribute__((aligned(16))) LinkRec {
    uint16_t to_lane, to_edge, foe_start, via2;     // via2 0xFFFF: none
    int16_t arr_idx;                                // approach register of this link (only foe targets have one)
    uint8_t tls, tls_pos;                           // tls 0xFF: uncontrolled
    uint8_t foe_cnt, flags;                         // flags bit0 minor, bit1 cont, bit2 to_lane is internal (= via1)
    uint8_t dest_k, pad;                            // lane index of the destination lane inside to_edge
    LaneRec dest;                                   // copy of lanes[to_lane]: one dependent gather less per hop
};
struct __attribute__((aligned(16))) FoeRec {
    int16_t arr_idx;
    uint8_t tls, tls_pos;
    uint16_t via1_cell0, via2_cell0;    // first cell of the foe's junction lanes (0xFFFF: none)
    uint8_t via1_nc, via2_nc, pad[6];   // number of cells of those lanes
};
#define CELL_LEN 64.0f
#define CELL_INV (1.0f / 64.0f)
struct __attribute__((aligned(16))) RStep {
    uint16_t edge, next_edge;       // next_edge 0xFFFF: last edge of the route
    uint32_t next_mask2, next_mask1;
    float tlsdist;
};
struct __attribute__((aligned(16))) RouteRec {
    uint32_t start;
    uint16_t depart_lane;
    int16_t depart_arr;         // insertion-candidate register of the departure lane
    uint16_t first_link;        // cache_link() of (departure lane, first route step), precomputed
    uint16_t depart_cell0;      // first list cell of the departure lane (a new vehicle always lands in it)
    float depart_len;           // length of the departure lane
};
#define LF_INTERNAL 1u
#define KF_MINOR 1u
#define KF_CONT 2u
#define KF_VIA1 4u

// tables used rarely (per signal, per tick by one wave, at load / observe): reached through one pointer
struct KCold {
    const int32_t *trip_depart, *trips_cum;
    const float *vtype_params;
    const uint8_t *tls8, *fix8;
    const int32_t *tls_nphase, *tls_ngreen, *tls_nlinks, *tls_state_off, *tls_dur_off, *tls_yel_off, *tls_dur, *tls_yellow;
    const int32_t *fix_nphase, *fix_state_off, *fix_dur_off, *fix_dur;
    const int16_t *lane_obs;
    const int32_t *obs_sig, *sig_obs_start, *mv_in_start, *mv_in_idx, *mv_out_start, *mv_out_idx, *pr_out_start, *pr_out_idx;
};
// tables of the per-vehicle, per-tick path: by value (SGPRs)
struct KTab {
    const LaneRec *lanes;
    const LinkRec *links;
    const FoeRec *foes;
    const RStep *rsteps;
    const uint32_t *route_mask2;
    const uint16_t *next_link;      // [n_route_steps][kmax]: choose_link() of a normal lane, 0xFFFF: none
    const RouteRec *routes;
    const uint16_t *trip_route;
    const uint8_t *trip_vtype;
    const KCold *cold;
    int32_t n_trips, tls_maxl, kmax;
    int32_t n_lanes, n_cells, n_signals, n_obs, n_vtypes, horizon, capacity, step_length, yellow_length, lmax, n_arr, n_dep;
};

// ------------------------------------------------------------------------------------------------ LDS view
struct __attribute__((aligned(8))) Node {
    float pos;
    uint16_t trip;      // 0xFFFF: free slot
    uint16_t nxt;       // next vehicle on the same lane (unordered), NIL terminated
};
struct Lds {
    struct Node *node;          // {pos, trip, next-in-lane}: one 8-byte LDS read per list step
    float *speed, *vnx, *tloss, *vtp;
    uint16_t *lane, *rq, *swait, *nlink;
    uint16_t *head;             // per-lane list heads (bit 15: the lane holds a moving vehicle)
    uint8_t *vt;
    int32_t *arr, *dep;         // link approach registers / departure-lane insertion candidates
    int32_t *agg_q, *agg_a, *agg_w, *agg_m;
    uint32_t *agg_s;
    int32_t *phase, *left, *nextp;
    uint8_t *tstate;            // current link states of every signal, [S][tls_maxl] (refreshed when a phase changes)
    int32_t *sc;        // scalars, see SC_*
};
#define SC_T 0
#define SC_NEXT 1
#define SC_HW 2
#define SC_HWNEW 3
#define SC_NPEND 4
#define SC_NLC 5
#define SC_STATS 6

__host__ __device__ inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }
__host__ __device__ inline size_t lds_scratch_bytes(int C, int n_obs) {       // vnx, later reused by the aggregates
    size_t a = (size_t)C * 4, b = align16((size_t)n_obs * 4) * 5;
    return a > b ? a : b;
}
// LDS layout.  The per-slot arrays, the scalars and the list heads come first: with the capacity a template
// parameter of the step kernel their offsets are compile-time constants (DS immediate offsets, no SGPR each);
// the arrays whose size depends on the scenario follow.
__host__ __device__ inline size_t lds_bytes_for(int C, int n_cells, int n_arr, int n_dep, int n_obs, int S, int n_vt, int tls_maxl) {
    size_t o = 0;
    o += (size_t)C * 8;                              // node {pos, trip, nxt}
    o += (size_t)C * 4 * 2;                          // speed tloss
    o += (size_t)C * 2 * 4;                          // lane rq swait nlink
    o += (size_t)C;                                  // vt
    o += align16((size_t)(SC_STATS + ST_N) * 4);     // scalars
    o += align16((size_t)(n_cells + 2) * 2);         // head per list cell (u16, CAS on the containing dword)
    o += align16(lds_scratch_bytes(C, n_obs));       // vnx | aggregates
    o += align16((size_t)n_vt * VT_COLS * 4);        // vtype table
    o += align16((size_t)n_arr * 4);                 // approach registers
    o += align16((size_t)n_dep * 4);                 // insertion candidates
    o += align16((size_t)S * 4) * 3 + align16((size_t)S * tls_maxl);   // tls phase/left/next + link states
    return o;
}
__device__ __forceinline__ void lds_carve(Lds &L, char *base, int C, int n_cells, int n_arr, int n_dep, int n_obs, int S, int n_vt, int tls_maxl) {
    size_t o = 0;
#define CARVE(field, type, bytes) L.field = (type *)(base + o); o += align16(bytes);
    CARVE(node, Node, (size_t)C * 8) CARVE(speed, float, (size_t)C * 4) CARVE(tloss, float, (size_t)C * 4)
    CARVE(lane, uint16_t, (size_t)C * 2)
    CARVE(rq, uint16_t, (size_t)C * 2) CARVE(swait, uint16_t, (size_t)C * 2) CARVE(nlink, uint16_t, (size_t)C * 2)
    CARVE(vt, uint8_t, (size_t)C)
    CARVE(sc, int32_t, (size_t)(SC_STATS + ST_N) * 4)
    CARVE(head, uint16_t, (size_t)(n_cells + 2) * 2)
    {   // the per-lane aggregates of the observe phase live where vnx was (dead by then)
        char *sb = base + o;
        L.vnx = (float *)sb;
        const size_t ab = align16((size_t)n_obs * 4);
        L.agg_q = (int32_t *)sb; L.agg_a = (int32_t *)(sb + ab); L.agg_w = (int32_t *)(sb + 2 * ab);
        L.agg_m = (int32_t *)(sb + 3 * ab); L.agg_s = (uint32_t *)(sb + 4 * ab);
        o += align16(lds_scratch_bytes(C, n_obs));
    }
    CARVE(vtp, float, (size_t)n_vt * VT_COLS * 4)
    CARVE(arr, int32_t, (size_t)n_arr * 4) CARVE(dep, int32_t, (size_t)n_dep * 4)
    CARVE(phase, int32_t, (size_t)S * 4) CARVE(left, int32_t, (size_t)S * 4) CARVE(nextp, int32_t, (size_t)S * 4)
    CARVE(tstate, uint8_t, (size_t)S * tls_maxl)
#undef CARVE
}

// ------------------------------------------------------------------------------------------------ helpers
__device__ __forceinline__ bool ahead_of(float pj, int kj, float pi, int ki) { return pj > pi || (pj == pi && kj < ki); }

__device__ __forceinline__ float speed_factor(const KParams &P, int env, int trip, const float *vt) {
    if (!P.speed_dev) return vt[VT_SF_MEAN];
    float s = 0.0f;
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) s += d_u01(d_hash(P.seed, (uint32_t)env, (uint32_t)trip, 0xFFFFFFFFu, i));
    float z = (s - 2.0f) * 1.7320508f;
    float f = vt[VT_SF_MEAN] + vt[VT_SF_DEV] * z;
    if (f < 0.2f) f = 0.2f;
    if (f > 2.0f) f = 2.0f;
    return f;
}

// push slot s on the list of `lane`; returns the previous head.  16-bit heads, exchanged with a CAS on the
// containing dword (LDS has no 16-bit atomics; a dword holds the heads of two neighbouring lanes)
__device__ __forceinline__ uint16_t list_push(uint16_t *head, int lane, int s, bool mover) {
    uint32_t *w = (uint32_t *)head + (lane >> 1);
    const int sh = (lane & 1) * 16;
    const uint32_t flag = mover ? 0x8000u : 0u;
    uint32_t old = *w, assumed;
    COUNT(9, 1);
    do {
        COUNT(10, 1);
        assumed = old;
        const uint32_t keep = (assumed >> sh) & 0x8000u;          // sticky mover flag of the lane
        old = atomicCAS(w, assumed, (assumed & ~(0xFFFFu << sh)) | (((uint32_t)s | keep | flag) << sh));
    } while (old != assumed);
    return (uint16_t)((old >> sh) & 0x7FFFu);
}
__device__ __forceinline__ void heads_clear(uint16_t *head, int n_lanes, int tid, int B) {
    for (int i = tid; i < (n_lanes + 2) / 2; i += B) ((uint32_t *)head)[i] = 0x7FFF7FFFu;
}

// the link a vehicle on lane `lane` (record LR) takes at route step rq; -1: none (last edge / wrong lane).  For a normal
// lane it is a function of (route step, lane index on the edge) only: rs_create tabulates it (prefer a destination
// lane from which the route goes on without a lane change, then one from which it still can, then any connection
// to the next edge), so the look-ahead does one gather per hop instead of a route-step fetch plus a scan of the links
__device__ __forceinline__ int choose_link(const KTab &T, const LaneRec &LR, int lane, int rq) {
    if (LR.link_cnt == 0) return -1;
    if (LR.flags & LF_INTERNAL) return LR.link_start;
    const int v = T.next_link[rq * T.kmax + (lane - (int)LR.edge_lane0)];
    return v == 0xFFFF ? -1 : v;
}

// value cached in L.nlink: the link index (0x7FFF: none) with bit 15 set when the link owns an approach register
#define NLINK_NONE 0x7FFF
#define NLINK_ARR 0x8000
__device__ __forceinline__ uint16_t cache_link(const KTab &T, const LaneRec &LR, int lane, int rq) {
    const int link = choose_link(T, LR, lane, rq);
    if (link < 0) return NLINK_NONE;
    return (uint16_t)(link | (T.links[link].arr_idx >= 0 ? NLINK_ARR : 0));
}

// A vehicle waiting for insertion keeps everything the insertion needs in its (otherwise unused) slot fields, so
// that the per-tick insertion phases touch LDS only:  pos = insertion position, swait = candidate register,
// nlink = first link, speed bits = departure lane << 16 | list cell, tloss bits = scheduled departure tick.
__device__ __forceinline__ void stash_pending(const KTab &T, Lds &L, int s, int k, const float *vt) {
    const RouteRec RR = T.routes[T.trip_route[k]];
    const float mypos = vt[VT_LENGTH] < RR.depart_len ? vt[VT_LENGTH] : RR.depart_len;
    L.node[s].pos = mypos;
    L.swait[s] = (uint16_t)RR.depart_arr;
    L.nlink[s] = RR.first_link;
    L.speed[s] = __int_as_float(((int)RR.depart_lane << 16) | (int)RR.depart_cell0);
    L.tloss[s] = __int_as_float(T.cold->trip_depart[k]);
}

__device__ __forceinline__ int tls_state(const KTab &T, const Lds &L, const KParams &P, int tls, int pos) {
    if (tls == 0xFF) return TLS_G;
    return L.tstate[tls * T.tls_maxl + pos];
}

__device__ __forceinline__ int lane_cells(const LaneRec &LR) { return (int)(LR.len * CELL_INV) + 1; }
__device__ __forceinline__ int cell_of(float pos, int ncell) { const int c = (int)(pos * CELL_INV); return c < ncell ? c : ncell - 1; }

// rear-most vehicle of a lane (min pos, ties -> larger trip index): the first non-empty cell holds it
__device__ __forceinline__ int rearmost(const Lds &L, const uint16_t *head, int cell0, int ncell) {
    COUNT(4, 1);
    for (int c = 0; c < ncell; ++c) {
        COUNT(5, 1);
        int s = head[cell0 + c] & 0x7FFF;
        if (s == NIL) continue;
        int best = NIL, bk = 0;
        float bp = 0.0f;
        while (s != NIL) {
            const Node nd = L.node[s];
            const int k = nd.trip;
            const float p = nd.pos;
            if (best == NIL || p < bp || (p == bp && k > bk)) { best = s; bk = k; bp = p; }
            COUNT(6, 1);
            s = nd.nxt;
        }
        return best;
    }
    return NIL;
}

// nearest vehicle ahead of (pos, k) on the lane: my own cell first, then the first non-empty cell further on
__device__ __forceinline__ int leader_of(const Lds &L, const uint16_t *head, int cell0, int ncell, float pos, int k, int self) {
    int Ld = NIL, Lk = 0;
    float Lp = 0.0f;
    int c = cell_of(pos, ncell);
    COUNT(0, 1);
    for (int s = head[cell0 + c] & 0x7FFF; s != NIL;) {
        const Node nd = L.node[s];
        const int cur = s;
        COUNT(1, 1);
        s = nd.nxt;
        if (cur == self) continue;
        if (ahead_of(nd.pos, nd.trip, pos, k) && (Ld == NIL || ahead_of(Lp, Lk, nd.pos, nd.trip))) { Ld = cur; Lk = nd.trip; Lp = nd.pos; }
    }
    for (c += 1; Ld == NIL && c < ncell; ++c) {
        COUNT(2, 1);
        for (int s = head[cell0 + c] & 0x7FFF; s != NIL;) {
            const Node nd = L.node[s];
            COUNT(3, 1);
            if (Ld == NIL || ahead_of(Lp, Lk, nd.pos, nd.trip)) { Ld = s; Lk = nd.trip; Lp = nd.pos; }
            s = nd.nxt;
        }
    }
    return Ld;
}

// nearest vehicles ahead of and behind (pos, k) on the lane
__device__ __forceinline__ void neighbours(const Lds &L, const uint16_t *head, int cell0, int ncell, float pos, int k, int self, int &lead, int &foll) {
    int Ld = NIL, Fd = NIL, Lk = 0, Fk = 0;
    float Lp = 0.0f, Fp = 0.0f;
    const int c0 = cell_of(pos, ncell);
    COUNT(7, 1);
    for (int s = head[cell0 + c0] & 0x7FFF; s != NIL;) {
        const Node nd = L.node[s];
        const int cur = s;
        COUNT(8, 1);
        s = nd.nxt;
        if (cur == self) continue;
        const int ks = nd.trip;
        const float ps = nd.pos;
        if (ahead_of(ps, ks, pos, k)) {
            if (Ld == NIL || ahead_of(Lp, Lk, ps, ks)) { Ld = cur; Lk = ks; Lp = ps; }
        } else {
            if (Fd == NIL || ahead_of(ps, ks, Fp, Fk)) { Fd = cur; Fk = ks; Fp = ps; }
        }
    }
    for (int c = c0 + 1; Ld == NIL && c < ncell; ++c)
        for (int s = head[cell0 + c] & 0x7FFF; s != NIL;) {
            const Node nd = L.node[s];
            COUNT(8, 1);
            if (Ld == NIL || ahead_of(Lp, Lk, nd.pos, nd.trip)) { Ld = s; Lk = nd.trip; Lp = nd.pos; }
            s = nd.nxt;
        }
    for (int c = c0 - 1; Fd == NIL && c >= 0; --c)
        for (int s = head[cell0 + c] & 0x7FFF; s != NIL;) {
            const Node nd = L.node[s];
            COUNT(8, 1);
            if (Fd == NIL || ahead_of(nd.pos, nd.trip, Fp, Fk)) { Fd = s; Fk = nd.trip; Fp = nd.pos; }
            s = nd.nxt;
        }
    lead = Ld; foll = Fd;
}

__device__ __forceinline__ bool cells_have_mover(const uint16_t *head, int cell0, int nc) {
    for (int c = 0; c < nc; ++c) { COUNT(13, 1); if (head[cell0 + c] & 0x8000) return true; }
    return false;
}

__device__ __forceinline__ bool foe_blocked(const KTab &T, const Lds &L, const uint16_t *head, const KParams &P, const LinkRec &K) {
    for (int i = K.foe_start; i < K.foe_start + K.foe_cnt; ++i) {
        const FoeRec F = T.foes[i];
        COUNT(12, 1);
        if (F.tls != 0xFF && tls_state(T, L, P, F.tls, F.tls_pos) == TLS_R) continue;
        if (F.arr_idx >= 0 && L.arr[F.arr_idx] < FOE_GAP_Q) return true;
        if (F.via1_cell0 != 0xFFFF && cells_have_mover(head, F.via1_cell0, F.via1_nc)) return true;   // a moving vehicle on
        if (F.via2_cell0 != 0xFFFF && cells_have_mover(head, F.via2_cell0, F.via2_nc)) return true;   // the foe's junction lanes
    }
    return false;
}

// copy the link states of signal s in phase ph into LDS (called by the thread that owns the signal)
__device__ __forceinline__ void tls_refresh(const KTab &T, Lds &L, const KParams &P, int s, int ph) {
    const uint8_t *src = (P.fixed_program ? T.cold->fix8 + T.cold->fix_state_off[s] : T.cold->tls8 + T.cold->tls_state_off[s]) + ph * T.cold->tls_nlinks[s];
    const int n = T.cold->tls_nlinks[s];
    for (int i = 0; i < n; ++i) L.tstate[s * T.tls_maxl + i] = src[i];
}

__device__ __forceinline__ void set_phase(const KTab &T, Lds &L, const KParams &P, int s, int ph) {
    if (ph < 0 || ph >= T.cold->tls_nphase[s]) return;
    L.phase[s] = ph;
    L.left[s] = T.cold->tls_dur[T.cold->tls_dur_off[s] + ph];
    tls_refresh(T, L, P, s, ph);
}

// strategic lane-change direction on an edge of n lanes for a vehicle on lane index kk whose route continues on the
// lanes of mask m2: towards the nearest of them (right on a tie), 0 when kk itself continues the route (or none does)
__device__ __forceinline__ int strategic_dir(uint32_t m2, int kk, int n) {
    if ((m2 >> kk) & 1u) return 0;
    int dl = 1000, dr = 1000;
    for (int j = kk + 1; j < n; ++j) if ((m2 >> j) & 1u) { dl = j - kk; break; }
    for (int j = kk - 1; j >= 0; --j) if ((m2 >> j) & 1u) { dr = kk - j; break; }
    if (dl == 1000 && dr == 1000) return 0;
    return (dr <= dl) ? -1 : +1;
}
// the vehicle on the lane with cells [cell0, cell0 + ncell) whose body overlaps the one at (pos, k) lengthwise (the
// nearer one ahead first), NIL: none
__device__ __forceinline__ int overlapping(const Lds &L, const uint16_t *head, int cell0, int ncell, float pos, int k, int self, float len_self) {
    int lead, foll;
    neighbours(L, head, cell0, ncell, pos, k, self, lead, foll);
    if (lead != NIL && L.node[lead].pos - L.vtp[L.vt[lead] * VT_COLS + VT_LENGTH] - pos < 0.0f) return lead;
    if (foll != NIL && pos - len_self - L.node[foll].pos < 0.0f) return foll;
    return NIL;
}

// Preparation of a tick for slot s (P2b + P3): a pending trip bids for its departure lane (lowest trip wins);
// a moving vehicle whose next link somebody may have to yield to registers its arrival time there.
__device__ __forceinline__ void tick_prepare(const KTab &T, Lds &L, const KParams &P, int s) {
    const int lane = L.lane[s];
    if (lane == LANE_PENDING) { atomicMin(&L.dep[L.swait[s]], (int)L.node[s].trip); return; }
    if (lane > LANE_PENDING || DIAG_SKIP(1)) return;
    const int nlk = L.nlink[s];
    if (!(nlk & NLINK_ARR)) return;         // nobody yields to my next link (or I have none)
    const float v = L.speed[s];
    if (v <= HALT_SPEED) return;
    const LinkRec K = T.links[nlk & 0x7FFF];
    const int st = tls_state(T, L, P, K.tls, K.tls_pos);
    if (st == TLS_R) return;
    const float dist = T.lanes[lane].len - L.node[s].pos;
    if (st == TLS_Y && dist >= d_brake_gap(v, L.vtp[L.vt[s] * VT_COLS + VT_DECEL])) return;
    const float ta = dist / (v > 1.0f ? v : 1.0f);
    const int q = ta * 10.0f >= 65000.0f ? 65000 : (int)(ta * 10.0f);
    atomicMin(&L.arr[K.arr_idx], q);
}

// ------------------------------------------------------------------------------------------------ the step kernel
// grid = n_envs workgroups (one environment each); blockDim.x = 64 * waves (<= 1024).
// __launch_bounds__(1024, 8): <= 64 VGPRs so that 32 waves (e.g. two 1024-thread workgroups) share a CU.
// CAP: the slot capacity as a compile-time constant (0: read it from the tables at run time)
template <int CAP>
__global__ void __launch_bounds__(1024, 8)
rs_step_kernel(KTab T, State G, Out O, KParams P, const int32_t *__restrict__ actions) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int env = blockIdx.x;
    if (env >= P.n_envs) return;
    const int tid = threadIdx.x, B = blockDim.x;
    const int C = CAP ? CAP : T.capacity, S = T.n_signals, NO = T.n_obs;
    const int genv = P.env_base + env;
    Lds L;
    lds_carve(L, smem, C, T.n_cells, T.n_arr, T.n_dep, NO, S, T.n_vtypes, T.tls_maxl);
    unsigned long long pt_ = 0;
#ifdef RS_BARWAIT
    const unsigned long long born_ = wall_clock64();
#endif
#define PROF_START() if (P.prof && tid == 0) pt_ = wall_clock64();
#define PROF_MARK(i_) if (P.prof && tid == 0) { unsigned long long n_ = wall_clock64(); atomicAdd(&P.prof[i_], n_ - pt_); pt_ = n_; }
    PROF_START()
    const size_t eo = (size_t)env * C;
    uint16_t *const hc = L.head;    // per-lane vehicle lists of the current state
    uint16_t *const hn = L.head;    // (one buffer: cleared between plan and move, rebuilt by the move)

    // ---- load the environment slab (once per env-step)
    if (tid < SC_STATS + ST_N) L.sc[tid] = tid < 3 ? G.env[env * 4 + tid] : 0;
    for (int i = tid; i < T.n_vtypes * VT_COLS; i += B) L.vtp[i] = T.cold->vtype_params[i];
    heads_clear(L.head, T.n_cells, tid, B);
    for (int i = tid; i < T.n_arr; i += B) L.arr[i] = ARR_NONE;
    for (int i = tid; i < T.n_dep; i += B) L.dep[i] = ARR_NONE;
    for (int i = tid; i < S; i += B) {
        int ph = G.tls[(env * S + i) * 3 + 0];
        L.phase[i] = ph;
        L.left[i] = G.tls[(env * S + i) * 3 + 1];
        L.nextp[i] = G.tls[(env * S + i) * 3 + 2];
        tls_refresh(T, L, P, i, ph);
    }
    SYNC();
    {
        const int hw0 = L.sc[SC_HW];
        int npend = 0;
        for (int s = tid; s < C; s += B) {
            uint16_t ln = LANE_NONE, tr = 0xFFFF;
            if (s < hw0) { ln = G.lane()[eo + s]; tr = G.trip()[eo + s]; }
            L.lane[s] = ln; L.node[s].trip = tr;
            if (ln != LANE_NONE) {
                const float sp = G.speed()[eo + s];
                L.node[s].pos = G.pos()[eo + s]; L.speed[s] = sp; L.swait[s] = G.swait()[eo + s]; L.tloss[s] = G.tloss()[eo + s];
                const int rq = (int)T.routes[T.trip_route[tr]].start + (int)G.cursor()[eo + s];
                L.rq[s] = (uint16_t)rq;
                L.vt[s] = T.trip_vtype[tr];
                if (ln != LANE_PENDING) {
                    const LaneRec LR0 = T.lanes[ln];
                    L.nlink[s] = cache_link(T, LR0, ln, rq);
                    L.node[s].nxt = list_push(hc, LR0.cell0 + cell_of(L.node[s].pos, lane_cells(LR0)), s, sp > HALT_SPEED);
                } else {
                    npend += 1;
                    stash_pending(T, L, s, tr, T.cold->vtype_params + T.trip_vtype[tr] * VT_COLS);
                }
            }
        }
        if (npend) atomicAdd(&L.sc[SC_NPEND], npend);
    }
    SYNC();
    PROF_MARK(0)

// TLS switch events at the beginning of a tick (P0), preceded by Signal.set_phase when the yellow ticks are over
#define TLS_BEGIN_OF_TICK(tick_)                                                                                   \
    for (int s_ = tid; s_ < S; s_ += B) {                                                                          \
        if (P.do_fsm && !P.fixed_program && (tick_) == T.yellow_length) set_phase(T, L, P, s_, L.nextp[s_]);       \
        int left_ = L.left[s_];                                                                                    \
        if (left_ == 0) {                                                                                          \
            const int32_t *dur_ = P.fixed_program ? T.cold->fix_dur + T.cold->fix_dur_off[s_] : T.cold->tls_dur + T.cold->tls_dur_off[s_]; \
            const int Pn_ = P.fixed_program ? T.cold->fix_nphase[s_] : T.cold->tls_nphase[s_];                                 \
            const int ph_ = (L.phase[s_] + 1) % Pn_;                                                               \
            left_ = dur_[ph_];                                                                                     \
            L.phase[s_] = ph_;                                                                                     \
            tls_refresh(T, L, P, s_, ph_);                                                                          \
        }                                                                                                          \
        L.left[s_] = left_ - 1;                                                                                    \
    }

// P2a for tick t_: departed trips take the lowest free slots in trip order (wave 0 only; base_hw_ = current hw)
#define ALLOCATE_SLOTS(t_)                                                                                         \
    if (tid < 64) {                                                                                                \
        const int hz_ = (t_) - 1 <= T.horizon ? (t_) - 1 : T.horizon;                                              \
        const int due_ = (t_) >= 1 ? T.cold->trips_cum[hz_] : 0;                                                         \
        const int nt_ = L.sc[SC_NEXT];                                                                             \
        const int m_ = due_ - nt_;                                                                                 \
        if (m_ > 0) {                                                                                              \
            int base_ = 0;                                                                                         \
            for (int c0 = 0; c0 < C && base_ < m_; c0 += 64) {                                                     \
                const int s_ = c0 + tid;                                                                           \
                const bool fr_ = L.lane[s_] == LANE_NONE;                                                          \
                const unsigned long long mask_ = __ballot(fr_);                                                    \
                const int rank_ = __popcll(mask_ & ((1ull << tid) - 1ull));                                        \
                if (fr_ && base_ + rank_ < m_) {                                                                   \
                    const int k_ = nt_ + base_ + rank_;                                                            \
                    const int v_ = T.trip_vtype[k_];                                                               \
                    L.node[s_].trip = (uint16_t)k_; L.lane[s_] = LANE_PENDING;                                          \
                    L.vt[s_] = (uint8_t)v_; L.rq[s_] = (uint16_t)T.routes[T.trip_route[k_]].start;                 \
                    stash_pending(T, L, s_, k_, T.cold->vtype_params + v_ * VT_COLS);                              \
                    G.sf()[eo + s_] = speed_factor(P, genv, k_, T.cold->vtype_params + v_ * VT_COLS);                      \
                    G.rwait()[eo + s_] = 0; G.owner()[eo + s_] = OWNER_NONE; G.depart()[eo + s_] = 0; G.accel()[eo + s_] = 0.0f; G.wtot()[eo + s_] = 0; \
                    atomicMax(&L.sc[SC_HW], s_ + 1);                                                               \
                }                                                                                                  \
                base_ += __popcll(mask_);                                                                          \
            }                                                                                                      \
            if (tid == 0) { const int got_ = base_ < m_ ? base_ : m_; L.sc[SC_NEXT] = nt_ + got_; L.sc[SC_NPEND] += got_; } \
        }                                                                                                          \
    }

    // ---- Signal.prep_phase for every signal (traffic_signal.py:176-184), then the preparation of tick 0
    if (P.do_fsm && !P.fixed_program) {
        for (int s = tid; s < S; s += B) {
            int a = actions[env * S + s], cur = L.phase[s], Gn = T.cold->tls_ngreen[s];
            if (a < 0 || a >= T.cold->tls_nphase[s]) { L.nextp[s] = cur; continue; }
            L.nextp[s] = a;
            if (cur != a && cur < Gn && a < Gn) {
                int y = T.cold->tls_yellow[T.cold->tls_yel_off[s] + cur * Gn + a];
                if (y >= 0) set_phase(T, L, P, s, y);
            }
        }
    }
    if (P.n_ticks > 0) {
        TLS_BEGIN_OF_TICK(0)
        ALLOCATE_SLOTS(L.sc[SC_T])
    }
    SYNC();
    PROF_MARK(1)
    // ---- A: insertion candidates and approach registration of the first tick (later ticks prepare the next
    //         one while they rebuild the lists, see F)
    // time, high-water mark and "somebody waits for insertion" of the tick about to run.  They are read while
    // nobody can change them (B decrements SC_NPEND): behind the barrier above / the one after E, and in front
    // of the barrier that ends the preparation sweep - `pending` guards a __syncthreads and must be block-uniform.
    int t = L.sc[SC_T], hw = L.sc[SC_HW];
    bool pending = L.sc[SC_NPEND] > 0;
    if (P.n_ticks > 0) {
        for (int s = tid; s < hw; s += B) tick_prepare(T, L, P, s);
        SYNC();
    }
    PROF_MARK(2)

    for (int tick = 0; tick < P.n_ticks; ++tick) {
        // ---- B: the candidate of each departure lane checks the space and inserts itself (P2c + P2d)
        if (pending) {
            for (int s = tid; s < hw; s += B) {
                if (L.lane[s] != LANE_PENDING) continue;
                const int k = L.node[s].trip;
                const int di = L.swait[s];
                if (L.dep[di] != k) continue;                   // lost (or the winner already cleared the register)
                const int packed = __float_as_int(L.speed[s]);
                const int dl = packed >> 16, cell = packed & 0xFFFF;
                const float *vt = L.vtp + L.vt[s] * VT_COLS;
                const float mypos = L.node[s].pos;
                bool ins = true;
                // only vehicles with pos < mypos + minGap + length can be in the way: they all sit in cell 0
                for (int o = hc[cell] & 0x7FFF; o != NIL; o = L.node[o].nxt) {
                    float back = L.node[o].pos - L.vtp[L.vt[o] * VT_COLS + VT_LENGTH];
                    if (back - mypos - vt[VT_MINGAP] < 0.0f) ins = false;
                }
                L.dep[di] = ARR_NONE;
                if (!ins) continue;
                const int sched = __float_as_int(L.tloss[s]);
                L.lane[s] = (uint16_t)dl; L.speed[s] = 0.0f; L.swait[s] = 0; L.tloss[s] = 0.0f;
                G.depart()[eo + s] = (uint16_t)t;
                L.node[s].nxt = list_push(hc, cell, s, false);
                atomicAdd(&L.sc[SC_STATS + ST_INSERTED], 1);
                atomicAdd(&L.sc[SC_STATS + ST_DEPDELAY], t - 1 - sched);
                atomicSub(&L.sc[SC_NPEND], 1);
            }
            SYNC();
            PROF_MARK(3)
        }
        // ---- C: plan (Krauss car-following + links)
        if (tid == 0) { L.sc[SC_HWNEW] = 0; L.sc[SC_NLC] = 0; }     // written again in D / E, behind barriers
        for (int s = tid; s < hw; s += B) {
            const int lane = L.lane[s];
            if (lane >= LANE_PENDING) continue;
            const int k = L.node[s].trip;
            COUNT(17, 1);
            const float *vt = L.vtp + L.vt[s] * VT_COLS;
            const float a = vt[VT_ACCEL], b = vt[VT_DECEL], tau = vt[VT_TAU], mingap = vt[VT_MINGAP];
            const float v = L.speed[s], x = L.node[s].pos;
            const float sf = G.sf()[eo + s];
            LaneRec LR = T.lanes[lane];
            int link = (int)(L.nlink[s] & 0x7FFF);
            float vfree = v + a;
            const float vl = LR.vmax * sf;
            if (vl < vfree) vfree = vl;
            if (vt[VT_MAXSPEED] < vfree) vfree = vt[VT_MAXSPEED];
            float vsafe = BIGF;
            // The look-ahead only FINDS what limits the vehicle (a leader, or a stop line = a standing leader of
            // zero length): the Krauss safe speed is evaluated once, by all lanes together, after the walk.
            float tgap = 0.0f, tvl = 0.0f, tbl = b;
            bool have = false;
            const int lead = leader_of(L, hc, LR.cell0, lane_cells(LR), x, k, s);
            bool found = false;
            if (lead != NIL) {
                const float *vo = L.vtp + L.vt[lead] * VT_COLS;
                tgap = L.node[lead].pos - vo[VT_LENGTH] - x - mingap;
                tvl = L.speed[lead]; tbl = vo[VT_DECEL];
                have = true;
                found = true;
            }
            const float look = d_brake_gap(vfree, b) + vfree * tau + mingap + 1.0f;
            float seen = LR.len - x;
            int rq = L.rq[s];
            if (link == NLINK_NONE) link = -1;
            if (!found && seen < look && !DIAG_SKIP(2)) {
                COUNT(18, 1);
                int cur_lane = lane;
                const float bgv = d_brake_gap(v, b);        // can I still stop in front of a red / yellow light?
                for (int hop = 0; hop < MAX_HOPS; ++hop) {
                    COUNT(11, 1);
                    const bool cur_int = (LR.flags & LF_INTERNAL) != 0;
                    if (hop > 0) link = choose_link(T, LR, cur_lane, rq);
                    bool stop_here = false;
                    LinkRec K;
                    if (link < 0) {
                        // last edge of the route: free run to its end; otherwise wrong lane: wait for a lane change
                        if (!cur_int && T.rsteps[rq].next_edge == 0xFFFF) break;
                        stop_here = true;
                    } else {
                        K = T.links[link];
                        const int st = tls_state(T, L, P, K.tls, K.tls_pos);
                        if (K.tls != 0xFF && (st == TLS_R || st == TLS_Y) && seen >= bgv) stop_here = true;
                        if (!stop_here && !(K.flags & KF_CONT) && K.foe_cnt > 0 && ((K.flags & KF_MINOR) || (K.tls != 0xFF && st == TLS_g))) {
                            if (!DIAG_SKIP(8) && foe_blocked(T, L, hc, P, K)) stop_here = true;
                        }
                    }
                    if (stop_here) {
                        const float g = seen - STOP_OFFSET;
                        tgap = g > 0.0f ? g : 0.0f; tvl = 0.0f; tbl = b;       // d_follow_speed(g, 0, b, b) == d_stop_speed(g, b)
                        have = true;
                        break;
                    }
                    LR = K.dest;
                    cur_lane = K.to_lane;
                    {   // slow down in time for a lower speed limit on the next lane
                        float vnl = LR.vmax * sf;
                        if (vnl < vfree) {
                            float vs = d_free_speed(seen, vnl, b);
                            if (vs < vsafe) vsafe = vs;
                        }
                    }
                    const int o = DIAG_SKIP(16) ? NIL : rearmost(L, hc, LR.cell0, lane_cells(LR));
                    if (o != NIL) {
                        const float *vo = L.vtp + L.vt[o] * VT_COLS;
                        tgap = seen + L.node[o].pos - vo[VT_LENGTH] - mingap;
                        tvl = L.speed[o]; tbl = vo[VT_DECEL];
                        have = true;
                        break;
                    }
                    if (!cur_int) rq += 1;
                    seen += LR.len;
                    if (!(seen < look) || DIAG_SKIP(32)) break;
                }
            }
            if (have) {
                const float vs = d_follow_speed(tgap, tvl, b, tbl, tau);
                if (vs < vsafe) vsafe = vs;
            }
            float vmin_n = v - b; if (vmin_n < 0.0f) vmin_n = 0.0f;
            float vmin_e = v - vt[VT_EMERGENCY]; if (vmin_e < 0.0f) vmin_e = 0.0f;
            const float lo = vsafe > vmin_e ? vsafe : vmin_e;
            const float vmin = vmin_n < lo ? vmin_n : lo;
            float vmax = vfree < vsafe ? vfree : vsafe;
            if (vmax < vmin) vmax = vmin;
            const float sigma = P.sigma >= 0.0f ? P.sigma : vt[VT_SIGMA];
            float vd = vmax;
            if (sigma > 0.0f) {
                const float r = d_u01(d_hash(P.seed, (uint32_t)genv, (uint32_t)k, (uint32_t)t, 0u));
                if (vd < a) vd -= sigma * vd * r; else vd -= sigma * a * r;
                if (vd < 0.0f) vd = 0.0f;
            }
            L.vnx[s] = vd > vmin ? vd : vmin;
        }
        SYNC();
        PROF_MARK(4)
        heads_clear(L.head, T.n_cells, tid, B);     // nobody reads the lists between plan and move
        for (int i = tid; i < T.n_arr; i += B) L.arr[i] = ARR_NONE;     // ... nor this tick's approach registrations
        SYNC();
        PROF_MARK(5)
        // ---- D: move; drop this tick's approach registrations; build the lists of the moved state
        {
            int active = 0, halted = 0, top = 0;
            for (int s = tid; s < hw; s += B) {
                int lane = L.lane[s];
                if (lane == LANE_NONE) continue;
                if (lane == LANE_PENDING) { top = s + 1; continue; }
                int link = (int)(L.nlink[s] & 0x7FFF);
                if (link == NLINK_NONE) link = -1;
                LaneRec LR = T.lanes[lane];
                const float sfv = G.sf()[eo + s];
                const float vn = L.vnx[s];
                const float vref = LR.vmax * sfv;
                if (tick == P.n_ticks - 1) G.accel()[eo + s] = vn - L.speed[s];
                L.speed[s] = vn;
                if (vn <= HALT_SPEED) {
                    int w = L.swait[s]; if (w < 65535) L.swait[s] = (uint16_t)(w + 1); halted += 1;
                    if (G.trip_log) { const int wt = G.wtot()[eo + s]; if (wt < 65535) G.wtot()[eo + s] = (uint16_t)(wt + 1); }
                } else L.swait[s] = 0;
                float tl = L.tloss[s];
                if (vref > 0.0f && vn < vref) { tl += (vref - vn) / vref; L.tloss[s] = tl; }
                float x = L.node[s].pos + vn;
                int rq = L.rq[s];
                bool arrived = false, moved = false;
                for (int it = 0; it < 16; ++it) {
                    if (!(x > LR.len)) break;
                    COUNT(14, 1);
                    const bool li = (LR.flags & LF_INTERNAL) != 0;
                    if (moved) link = choose_link(T, LR, lane, rq);
                    if (link < 0) {
                        if (!li && T.rsteps[rq].next_edge == 0xFFFF) arrived = true; else x = LR.len;
                        break;
                    }
                    x -= LR.len;
                    if (!li) rq += 1;
                    {
                        const LinkRec Km = T.links[link];
                        lane = Km.to_lane;
                        LR = Km.dest;
                    }
                    moved = true;
                }
                if (arrived) {
                    const int ktrip = L.node[s].trip;
                    L.lane[s] = LANE_NONE; L.node[s].trip = 0xFFFF;
                    G.owner()[eo + s] = OWNER_NONE; G.rwait()[eo + s] = 0;
                    atomicAdd(&L.sc[SC_STATS + ST_ARRIVED], 1);
                    atomicAdd(&L.sc[SC_STATS + ST_DURATION], t + 1 - (int)G.depart()[eo + s]);
                    atomicAdd(&L.sc[SC_STATS + ST_TLOSS], (int)(tl * 1024.0f + 0.5f));
                    if (G.trip_log) {
                        int32_t *r = G.trip_log + ((size_t)env * T.n_trips + ktrip) * 4;
                        r[0] = (int)G.depart()[eo + s]; r[1] = t + 1; r[2] = (int)(tl * 1024.0f + 0.5f); r[3] = (int)G.wtot()[eo + s];
                    }
                } else {
                    L.node[s].pos = x;
                    if (moved) {
                        L.lane[s] = (uint16_t)lane; L.rq[s] = (uint16_t)rq;
                        L.nlink[s] = cache_link(T, LR, lane, rq);
                    }
                    active += 1;
                    top = s + 1;
                    L.node[s].nxt = list_push(hn, LR.cell0 + cell_of(x, lane_cells(LR)), s, vn > HALT_SPEED);
                }
            }
            if (active) atomicAdd(&L.sc[SC_STATS + ST_ACTIVE_TICKS], active);
            if (halted) atomicAdd(&L.sc[SC_STATS + ST_WAITING], halted);
            if (top) atomicMax(&L.sc[SC_HWNEW], top);
        }
        SYNC();
        PROF_MARK(6)
        // ---- E: lane-change decisions on the moved state (all changes of a tick go the same way: left on even
        //         ticks); the next tick's TLS events and slot allocation are prepared in the same phase
        const int hw2 = L.sc[SC_HWNEW];
        const int dir_allowed = (t & 1) ? -1 : +1;
        for (int s = tid; s < hw2; s += B) {
            int target = -1;
            const int lane = L.lane[s];
            if (lane < LANE_PENDING && !DIAG_SKIP(4)) {
                const LaneRec LR = T.lanes[lane];
                const int n = LR.flags >> 2;
                const int l0 = LR.edge_lane0;
                const int kk = lane - l0;
                const int tk = kk + dir_allowed;
                if (!(LR.flags & LF_INTERNAL) && n >= 2) {
                    const int k = L.node[s].trip;
                    const uint32_t m2 = T.route_mask2[L.rq[s]];
                    const float *vt = L.vtp + L.vt[s] * VT_COLS;
                    const float x = L.node[s].pos, v = L.speed[s];
                    const float lane_len = LR.len;
                    const int nc = lane_cells(LR);
                    const int sdir = strategic_dir(m2, kk, n);      // 0: my lane continues the route
                    if (tk >= 0 && tk < n) {
                        const int tl = l0 + tk;
                        const int tcell0 = (int)LR.cell0 + dir_allowed * nc;     // lanes of an edge own consecutive cell blocks
                        int want = 0;
                        int lead_t = NIL, foll_t = NIL;
                        if (!((m2 >> kk) & 1u)) {
                            want = (sdir == dir_allowed) ? 2 : 0;
                            if (want) neighbours(L, hn, tcell0, nc, x, k, s, lead_t, foll_t);
                        } else if (((m2 >> tk) & 1u) && ((((uint32_t)t >> 1) + (uint32_t)k) & 3u) == 0u) {
                            const int lead_c = leader_of(L, hn, LR.cell0, nc, x, k, s);
                            if (lead_c != NIL) {
                                neighbours(L, hn, tcell0, nc, x, k, s, lead_t, foll_t);
                                float gcur = L.node[lead_c].pos - L.vtp[L.vt[lead_c] * VT_COLS + VT_LENGTH] - x;
                                float gtgt = BIGF;
                                if (lead_t != NIL) gtgt = L.node[lead_t].pos - L.vtp[L.vt[lead_t] * VT_COLS + VT_LENGTH] - x;
                                if (gcur < v * 3.0f + 15.0f && gtgt > gcur + SG_ADVANTAGE) want = 1;
                            }
                        }
                        if (want) {
                            COUNT(16, 1);
                            const bool urgent = want == 2 && (lane_len - x) <= URGENT_DIST;
                            bool safe = true;
                            if (lead_t != NIL) {
                                const float *vo = L.vtp + L.vt[lead_t] * VT_COLS;
                                float gap = L.node[lead_t].pos - vo[VT_LENGTH] - x - (urgent ? 0.0f : vt[VT_MINGAP]);
                                float dec = urgent ? vt[VT_EMERGENCY] : vt[VT_DECEL];
                                float vb = v - dec; if (vb < 0.0f) vb = 0.0f;
                                if (gap < 0.0f || vb > d_follow_speed(gap, L.speed[lead_t], vt[VT_DECEL], vo[VT_DECEL], vt[VT_TAU])) safe = false;
                            }
                            if (safe && foll_t != NIL) {
                                const float *vo = L.vtp + L.vt[foll_t] * VT_COLS;
                                float gap = x - vt[VT_LENGTH] - L.node[foll_t].pos - (urgent ? 0.0f : vo[VT_MINGAP]);
                                float dec = urgent ? vo[VT_EMERGENCY] : vo[VT_DECEL];
                                float vb = L.speed[foll_t] - dec; if (vb < 0.0f) vb = 0.0f;
                                if (gap < 0.0f || vb > d_follow_speed(gap, v, vo[VT_DECEL], vt[VT_DECEL], vo[VT_TAU])) safe = false;
                            }
                            if (safe) target = tl;
                        }
                    }
                    // Mutual block: two vehicles that have stood side by side near the end of their lanes for SWAP_WAIT
                    // seconds, each in the lane the other one needs, can never find a gap: they trade places.  The test is symmetric, so both
                    // threads reach the same verdict from the moved state (whichever way this tick's changes go).
                    if (sdir != 0 && (t % SWAP_EVERY) == 0 && v <= HALT_SPEED && L.swait[s] >= SWAP_WAIT && (lane_len - x) <= URGENT_DIST) {
                        const int b = overlapping(L, hn, (int)LR.cell0 + sdir * nc, nc, x, k, s, vt[VT_LENGTH]);
                        if (b != NIL && L.speed[b] <= HALT_SPEED && L.swait[b] >= SWAP_WAIT) {
                            const Node nb = L.node[b];
                            if (T.lanes[lane + sdir].len - nb.pos <= URGENT_DIST &&
                                strategic_dir(T.route_mask2[L.rq[b]], kk + sdir, n) == -sdir &&
                                overlapping(L, hn, LR.cell0, nc, nb.pos, nb.trip, b, L.vtp[L.vt[b] * VT_COLS + VT_LENGTH]) == s)
                                target = lane + sdir;
                        }
                    }
                }
            }
            L.vnx[s] = __int_as_float(target);
            if (target >= 0) L.sc[SC_NLC] = 1;
        }
        if (tid == 0) { L.sc[SC_T] = t + 1; L.sc[SC_HW] = hw2; L.sc[SC_STATS + ST_TICKS] += 1; }
        if (tick + 1 < P.n_ticks) {
            TLS_BEGIN_OF_TICK(tick + 1)
            ALLOCATE_SLOTS(t + 1)       // wave 0; tid 0 has just published hw2 (same wave, program order)
        }
        SYNC();
        PROF_MARK(7)
        // ---- F: when somebody changes lane: apply, rebuild the lists; in the same sweep (or on its own when
        //         nobody did) the next tick's insertion bids and approach registrations (P2b + P3)
        const bool more = tick + 1 < P.n_ticks;
        const int hwn = L.sc[SC_HW];        // hw2 + the slots allocated for the next tick
        const int tn = L.sc[SC_T];
        const bool pn = L.sc[SC_NPEND] > 0;
        if (L.sc[SC_NLC]) {
            heads_clear(hn, T.n_cells, tid, B);
            for (int s = tid; s < hw2; s += B) {
                const int target = __float_as_int(L.vnx[s]);
                if (L.lane[s] < LANE_PENDING && target >= 0) {
                    L.lane[s] = (uint16_t)target;
                    L.nlink[s] = cache_link(T, T.lanes[target], target, L.rq[s]);
                }
            }
            SYNC();
            for (int s = tid; s < hwn; s += B) {
                const int ln = L.lane[s];
                if (ln < LANE_PENDING) {
                    const LaneRec LRn = T.lanes[ln];
                    L.node[s].nxt = list_push(hn, LRn.cell0 + cell_of(L.node[s].pos, lane_cells(LRn)), s, L.speed[s] > HALT_SPEED);
                }
                if (more) tick_prepare(T, L, P, s);
            }
            SYNC();
        } else if (more) {
            for (int s = tid; s < hwn; s += B) tick_prepare(T, L, P, s);
            SYNC();
        }
        t = tn; hw = hwn; pending = pn;
        PROF_MARK(8)
    }
#undef TLS_BEGIN_OF_TICK
#undef ALLOCATE_SLOTS

    // ---- Signal.observe for every signal (traffic_signal.py:189-247)
    for (int i = tid; i < NO; i += B) { L.agg_q[i] = 0; L.agg_a[i] = 0; L.agg_w[i] = 0; L.agg_m[i] = 0; L.agg_s[i] = 0; }
    SYNC();
    const int hwf = L.sc[SC_HW];
    {
        const int hw0 = G.env[env * 4 + 2];
        const int top = hwf > hw0 ? hwf : hw0;
        int act = 0, pend = 0;
        for (int s = tid; s < top; s += B) {
            const int lane = L.lane[s];
            G.lane()[eo + s] = (uint16_t)lane; G.trip()[eo + s] = L.node[s].trip;
            if (lane == LANE_NONE) continue;
            // store the slab back (once per env-step)
            const int rq = L.rq[s];
            if (lane == LANE_PENDING) {     // the slot fields hold the insertion stash; the state of a waiting vehicle is all zero
                G.pos()[eo + s] = 0.0f; G.speed()[eo + s] = 0.0f; G.swait()[eo + s] = 0; G.tloss()[eo + s] = 0.0f; G.cursor()[eo + s] = 0;
                pend += 1;
                continue;
            }
            G.pos()[eo + s] = L.node[s].pos; G.speed()[eo + s] = L.speed[s]; G.swait()[eo + s] = L.swait[s]; G.tloss()[eo + s] = L.tloss[s];
            G.cursor()[eo + s] = (uint16_t)(rq - (int)T.routes[T.trip_route[L.node[s].trip]].start);
            act += 1;
            const LaneRec LR = T.lanes[lane];
            const int oi = T.cold->lane_obs[lane];
            bool detect = false;
            if (oi >= 0) {
                float d = (LR.len - L.node[s].pos) + T.rsteps[rq].tlsdist;
                detect = d <= P.max_distance;
            }
            if (!detect) { G.owner()[eo + s] = OWNER_NONE; G.rwait()[eo + s] = 0; continue; }
            const int sig = T.cold->obs_sig[oi];
            int rw = G.rwait()[eo + s];
            if (G.owner()[eo + s] != (uint8_t)sig) rw = 0;
            if (rw > 0) { rw += T.step_length; if (rw > 65535) rw = 65535; }
            else if (L.swait[s] > 0) rw = L.swait[s];
            G.rwait()[eo + s] = (uint16_t)rw;
            G.owner()[eo + s] = (uint8_t)sig;
            if (rw > 0) { atomicAdd(&L.agg_q[oi], 1); atomicAdd(&L.agg_w[oi], rw); atomicMax(&L.agg_m[oi], rw); }
            else atomicAdd(&L.agg_a[oi], 1);
            atomicAdd(&L.agg_s[oi], (uint32_t)(L.speed[s] * 65536.0f + 0.5f));
        }
        if (act) atomicAdd(&L.sc[SC_STATS + ST_ACTIVE], act);
        if (pend) atomicAdd(&L.sc[SC_STATS + ST_PENDING], pend);
    }
    SYNC();
    PROF_MARK(9)
    // per observed lane rows, written as flat coalesced streams (element i of [n_obs][5] / [S][Lmax][5])
    for (int i = tid; i < NO * 5; i += B) {
        const int oi = i / 5, c = i - oi * 5;
        const int sg = T.cold->obs_sig[oi];
        const float sp = (float)L.agg_s[oi] * (1.0f / 65536.0f);
        float raw, nrm;
        if (c == 0) { raw = (float)L.agg_q[oi]; nrm = (oi - T.cold->sig_obs_start[sg]) == L.phase[sg] ? 1.0f : 0.0f; }
        else if (c == 1) { raw = (float)L.agg_a[oi]; nrm = raw / 28.0f; }
        else if (c == 2) { raw = (float)L.agg_w[oi]; nrm = raw / 28.0f; }
        else if (c == 3) { raw = (float)L.agg_m[oi]; nrm = (float)L.agg_q[oi] / 28.0f; }
        else { raw = sp; nrm = sp / 20.0f / 28.0f; }
        O.lane_agg()[(size_t)env * NO * 5 + i] = raw;         // queue, approach, total_wait, max_wait, speed_sum
        O.drq_norm()[(size_t)env * NO * 5 + i] = nrm;         // one-hot(lane position == phase), approach, wait, queue, speed
    }
    for (int i = tid; i < S * T.lmax * 5; i += B) {
        const int sg = i / (T.lmax * 5), r = i - sg * T.lmax * 5;
        const int l = r / 5, c = r - l * 5;
        const int o0 = T.cold->sig_obs_start[sg];
        float nrm = 0.0f;                                    // zero padding beyond the signal's lanes
        if (l < T.cold->sig_obs_start[sg + 1] - o0) {
            const int oi = o0 + l;
            if (c == 0) nrm = l == L.phase[sg] ? 1.0f : 0.0f;
            else if (c == 1) nrm = (float)L.agg_a[oi] / 28.0f;
            else if (c == 2) nrm = (float)L.agg_w[oi] / 28.0f;
            else if (c == 3) nrm = (float)L.agg_q[oi] / 28.0f;
            else nrm = (float)L.agg_s[oi] * (1.0f / 65536.0f) / 20.0f / 28.0f;
        }
        O.drq_f16()[(size_t)env * S * T.lmax * 5 + i] = __float2half(nrm);
    }
    // states.mplight / states.wave: one thread per (signal, movement)
    for (int i = tid; i < S * 12; i += B) {
        const int sg = i / 12, m = i - sg * 12;
        int q = 0, wv = 0;
        for (int j = T.cold->mv_in_start[i]; j < T.cold->mv_in_start[i + 1]; ++j) {
            const int oi = T.cold->mv_in_idx[j];
            q += L.agg_q[oi]; wv += L.agg_q[oi] + L.agg_a[oi];
        }
        for (int j = T.cold->mv_out_start[i]; j < T.cold->mv_out_start[i + 1]; ++j) q -= L.agg_q[T.cold->mv_out_idx[j]];
        const size_t so = (size_t)env * S + sg;
        O.mplight()[so * 13 + 1 + m] = q;
        O.wave()[so * 12 + m] = wv;
    }
    // per signal: phase, rewards, metrics
    for (int sg = tid; sg < S; sg += B) {
        const int ph = L.phase[sg];
        const int o0 = T.cold->sig_obs_start[sg], o1 = T.cold->sig_obs_start[sg + 1];
        int tw = 0, tq = 0, mq = 0;
        for (int oi = o0; oi < o1; ++oi) { tw += L.agg_w[oi]; int qq = L.agg_q[oi]; tq += qq; if (qq > mq) mq = qq; }
        const size_t so = (size_t)env * S + sg;
        O.phase()[so] = ph; O.queue_sum()[so] = tq; O.queue_max()[so] = mq;
        O.wait()[so] = -(float)tw;
        const float wn = -(float)tw / 224.0f;
        O.wait_norm()[so] = wn < -4.0f ? -4.0f : (wn > 4.0f ? 4.0f : wn);
        int pr = tq;
        for (int i = T.cold->pr_out_start[sg]; i < T.cold->pr_out_start[sg + 1]; ++i) pr -= L.agg_q[T.cold->pr_out_idx[i]];
        O.pressure()[so] = -pr;
        O.mplight()[so * 13] = ph;
        G.tls[(env * S + sg) * 3 + 0] = ph;
        G.tls[(env * S + sg) * 3 + 1] = L.left[sg];
        G.tls[(env * S + sg) * 3 + 2] = L.nextp[sg];
    }
    SYNC();
    PROF_MARK(10)
    if (tid < 3) G.env[env * 4 + tid] = L.sc[tid];
    if (tid < ST_N) {
        long long *st = G.stats + (size_t)env * ST_N;
        if (tid == ST_ACTIVE || tid == ST_PENDING) st[tid] = L.sc[SC_STATS + tid];
        else st[tid] += L.sc[SC_STATS + tid];
    }
#ifdef RS_BARWAIT
    if ((tid & 63) == 0) atomicAdd(&g_bar[1], wall_clock64() - born_);
#endif
}

// reset every environment: no vehicles, TLS programs freshly installed (Signal.__init__, traffic_signal.py:93-100)
extern "C" __global__ void rs_reset_kernel(Tab T, State G, KParams P) {
    const int env = blockIdx.x;
    const int C = T.capacity, S = T.n_signals;
    const size_t eo = (size_t)env * C;
    for (int s = threadIdx.x; s < C; s += blockDim.x) {
        G.lane()[eo + s] = LANE_NONE; G.trip()[eo + s] = 0xFFFF; G.owner()[eo + s] = OWNER_NONE;
        G.rwait()[eo + s] = 0; G.swait()[eo + s] = 0; G.cursor()[eo + s] = 0; G.depart()[eo + s] = 0; G.wtot()[eo + s] = 0;
        G.pos()[eo + s] = 0.0f; G.speed()[eo + s] = 0.0f; G.accel()[eo + s] = 0.0f; G.tloss()[eo + s] = 0.0f; G.sf()[eo + s] = 1.0f;
    }
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
        int ph, left;
        if (P.fixed_program) { ph = T.fix_init_phase[s]; left = T.fix_init_left[s]; }
        else { ph = T.tls_init_phase[s]; left = T.tls_dur[T.tls_dur_off[s] + ph]; }
        G.tls[(env * S + s) * 3 + 0] = ph; G.tls[(env * S + s) * 3 + 1] = left; G.tls[(env * S + s) * 3 + 2] = 0;
    }
    if (threadIdx.x < 4) G.env[env * 4 + threadIdx.x] = 0;
    if (threadIdx.x < ST_N) G.stats[(size_t)env * ST_N + threadIdx.x] = 0;
    if (G.trip_log)
        for (int i = threadIdx.x; i < T.n_trips * 4; i += blockDim.x) G.trip_log[(size_t)env * T.n_trips * 4 + i] = 0;
}

// ------------------------------------------------------------------------------------------------ static agents
// STOCHASTIC (agents/stochastic.py:17-18): uniform green index per (env, signal, step)
extern "C" __global__ void rs_act_random_kernel(Tab T, KParams P, uint32_t step_key, int32_t *actions) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int S = T.n_signals;
    if (i >= P.n_envs * S) return;
    int env = i / S, s = i - env * S;
    uint32_t h = d_hash(P.seed ^ 0xA5A5A5A5u, (uint32_t)(P.env_base + env), (uint32_t)s, step_key, 7u);
    actions[i] = (int32_t)(h % (uint32_t)T.tls_ngreen[s]);
}
// MAXWAVE / MAXPRESSURE (agents/maxwave.py:18-38, maxpressure.py:13-18): first maximum over the valid
// phase pairs (in the reference's iteration order) of obs[pair0] + obs[pair1]
extern "C" __global__ void rs_act_maxwave_kernel(Tab T, KParams P, const int32_t *pairs, int n_pairs,
                                                 const int32_t *valid, const int32_t *order, int use_pressure,
                                                 const int32_t *mplight, const int32_t *wave, int32_t *actions) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int S = T.n_signals;
    if (i >= P.n_envs * S) return;
    int s = i % S;
    const int32_t *obs = use_pressure ? mplight + (size_t)i * 13 + 1 : wave + (size_t)i * 12;
    bool have = false;
    int best = 0, best_act = 0;
    for (int j = 0; j < n_pairs; ++j) {
        int p = order[s * n_pairs + j];     // the reference walks valid_acts in dict order; ties keep the first
        if (p < 0) break;
        int act = valid[s * n_pairs + p];
        if (act < 0) continue;
        int press = obs[pairs[p * 2]] + obs[pairs[p * 2 + 1]];
        if (!have || press > best) { have = true; best = press; best_act = act; }
    }
    actions[i] = best_act;
}

