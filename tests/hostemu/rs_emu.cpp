// rs_emu.cpp -- HOST EMULATION of the step kernel (TEST INFRASTRUCTURE, never shipped, never loaded by resco_amd/).
//
// Compiles resco_amd/csrc/resco_step.h -- the very source of the HIP kernel -- with g++ and runs the "threads" of a
// workgroup one after the other, phase by phase (HostExec::phase), in a selectable order (ascending, descending,
// shuffled).  The CPU tests compare it with the oracle bit for bit: that pins the kernel's LOGIC (and its independence
// of the thread order inside a phase) without a GPU; the -m gpu tests then pin the real thing.  The library exports the
// subset of the C ABI of include/resco_sim.h the tests need, so tests/hostemu/emu.py can reuse the ctypes wrapper.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#include "resco_sim.h"

#define RS_DEV static inline
#define RS_HD
#define RS_MEM inline
#define RS_CARVE static inline
static char *g_smem = nullptr;
#define RS_SMEM g_smem
// an invariant of the kernel source the emulation checks (the device build compiles it out)
#define RS_EMU_CHECK_MAIL 1
#define RS_ASSERT(c) if (!(c)) { fprintf(stderr, "rs_emu: invariant violated: %s (resco_step.h:%d)\n", #c, __LINE__); abort(); }
static inline void rs_atomic_min(int32_t *p, int32_t v) { if (v < *p) *p = v; }
static inline void rs_atomic_min(uint32_t *p, uint32_t v) { if (v < *p) *p = v; }
static inline void rs_atomic_max(int32_t *p, int32_t v) { if (v > *p) *p = v; }
static inline void rs_atomic_add(int32_t *p, int32_t v) { *p += v; }
static inline int32_t rs_atomic_fetch_add(int32_t *p, int32_t v) { const int32_t o = *p; *p += v; return o; }
static inline int32_t rs_wave_ticket(int32_t *p) { return (*p)++; }
static inline void rs_wave_add(int32_t *p, int32_t v) { *p += v; }
static inline void rs_wave_max(int32_t *p, int32_t v) { if (v > *p) *p = v; }
static inline void rs_atomic_or(uint32_t *p, uint32_t v) { *p |= v; }
static inline uint32_t rs_atomic_fetch_or(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p |= v; return o; }
static inline void rs_atomic_and(uint32_t *p, uint32_t v) { *p &= v; }
static inline uint32_t rs_atomic_cas(uint32_t *p, uint32_t cmp, uint32_t v) { uint32_t o = *p; if (o == cmp) *p = v; return o; }
static inline int rs_ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int rs_clzll(unsigned long long x) { return __builtin_clzll(x); }
static inline int rs_ffs(uint32_t x) { return __builtin_ffs((int)x); }
static inline int rs_popc(uint32_t x) { return __builtin_popcount(x); }
static inline float rs_int_as_float(int x) { float f; memcpy(&f, &x, 4); return f; }
static inline int rs_float_as_int(float x) { int i; memcpy(&i, &x, 4); return i; }
// float -> IEEE half bits, round to nearest even (what __float2half does)
static inline uint16_t rs_f2h(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    int32_t e = (int32_t)((x >> 23) & 0xFF) - 127 + 15;
    uint32_t m = x & 0x7FFFFFu;
    if (((x >> 23) & 0xFF) == 0xFF) return (uint16_t)(sign | 0x7C00u | (m ? 0x200u : 0u));
    if (e >= 31) return (uint16_t)(sign | 0x7C00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        m |= 0x800000u;
        const int shift = 14 - e;
        uint32_t r = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1u))) r += 1;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ((uint32_t)e << 10) | (m >> 13);
    const uint32_t rem = m & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r += 1;
    return (uint16_t)(sign | r);
}

#ifdef RS_EMU_DEBUG
#include <stdio.h>
static int rs_dbg_slot = getenv("RS_DBG_SLOT") ? atoi(getenv("RS_DBG_SLOT")) : -1;
static int rs_dbg_t = getenv("RS_DBG_T") ? atoi(getenv("RS_DBG_T")) : -1;
static long rs_dbg_chain = 0;         // chain-walk steps of the current phase (reset by HostExec::phase)
#endif
#include "resco_step.h"

struct HostExec : ExecInline {
    unsigned long long role_begin() const { return 0ull; }
    void role_end(int, unsigned long long) const {}
    int wave_of(int tid) const { return tid >> 6; }
    // chunk of work for wave w on its i-th call in a phase (n waves): the GPU hands chunks out dynamically; here a fixed
    // assignment that depends on the thread order under test -- every chunk goes to exactly one wave either way
    int next_chunk(int32_t *, int w, int i, int n) const { return i * n + (order == 0 ? w : (order == 1 ? n - 1 - w : (w + n / 2 + 1) % n)); }
    int B;
    int order;          // 0 ascending, 1 descending, 2 shuffled (a different permutation in every phase)
    uint32_t rng = 12345u;
    template <class F> void phase(int id, F f) {
#ifdef RS_EMU_DEBUG
        if (getenv("RS_DBG_PHASES")) { printf("phase %d\n", id); fflush(stdout); }
        rs_dbg_chain = 0;
#endif
        if (order == 0) for (int t = 0; t < B; ++t) f(t);
        else if (order == 1) for (int t = B - 1; t >= 0; --t) f(t);
        else {
            std::vector<int> p((size_t)B);
            for (int i = 0; i < B; ++i) p[i] = i;
            for (int i = B - 1; i > 0; --i) { rng = rng * 1664525u + 1013904223u; std::swap(p[i], p[(rng >> 8) % (uint32_t)(i + 1)]); }
            for (int t : p) f(t);
        }
    }
};

struct rs_sim {
    PackedTables PT;
    KTab K{};
    State G{};
    Out O{};
    KParams P{};
    Lds L{};
    int n_envs = 0, block = 0, order = 0, ratio = 1;
    size_t lds = 0;
    std::vector<char> slab, outb, smem;
    std::vector<int32_t> env, tls, actions, trip_log, pairs, valid, ordr;
    uint32_t out_mask = OUT_ALL;
    std::vector<long long> stats;
    std::vector<uint16_t> dep_next;
    std::vector<uint32_t> mail;
    std::vector<float> route_cont, vtype_params;
    std::vector<std::vector<int32_t>> keep;
    struct Buf { void *ptr; int64_t shape[4]; int ndim; int dtype; size_t bytes; };
    Buf bufs[RS_BUF_COUNT]{};
    std::string err;
    int n_pairs = 0;
};
static std::string g_err;
static const size_t kDtypeSize[] = {4, 4, 2, 1, 2, 8, 4};
static void set_buf(rs_sim *h, int which, void *ptr, int dtype, int ndim, int64_t a, int64_t b = 1, int64_t c = 1, int64_t d = 1) {
    auto &B = h->bufs[which];
    B.ptr = ptr; B.dtype = dtype; B.ndim = ndim;
    B.shape[0] = a; B.shape[1] = b; B.shape[2] = c; B.shape[3] = d;
    B.bytes = (size_t)(a * b * c * d) * kDtypeSize[dtype];
}
static const int32_t *keep_i32(rs_sim *h, const int32_t *src, size_t n) {
    h->keep.emplace_back(src, src + (n ? n : 0));
    if (h->keep.back().empty()) h->keep.back().push_back(0);
    return h->keep.back().data();
}

static void run_step(rs_sim *h, int n_ticks, int do_fsm, int do_observe = 1) {
    KParams P = h->P;
    P.n_ticks = n_ticks; P.do_fsm = do_fsm; P.do_observe = do_observe; P.out_mask = h->out_mask; P.prof = nullptr;
    for (int env = 0; env < h->n_envs; ++env) {
        std::fill(h->smem.begin(), h->smem.end(), (char)0xA5);      // LDS is not zero on the GPU either
        g_smem = h->smem.data();
        HostExec ex;
        ex.B = h->block; ex.order = h->order;
        ex.rng = 777u + (uint32_t)env;
        switch (h->K.capacity) {
            case 128: rs_step_body<128>(ex, h->L, h->K, h->G, h->O, P, h->actions.data(), env); break;
            case 256: rs_step_body<256>(ex, h->L, h->K, h->G, h->O, P, h->actions.data(), env); break;
            case 896: rs_step_body<896>(ex, h->L, h->K, h->G, h->O, P, h->actions.data(), env); break;
            case 1024: rs_step_body<1024>(ex, h->L, h->K, h->G, h->O, P, h->actions.data(), env); break;
            default: rs_step_body<0>(ex, h->L, h->K, h->G, h->O, P, h->actions.data(), env); break;
        }
    }
}

extern "C" {
// block_threads < 0 selects thread order: -1 descending, -2 shuffled (with one thread per slot); order > 0 as given
int rs_create(const rs_scenario *sc, const rs_params *p, int32_t n_envs, int32_t env_base, int32_t device_id, int32_t block_threads, rs_handle *out) {
    rs_sim *h = new rs_sim();
    {   // (the library's choice of the grid cell length, resco_sim.hip)
        PackedTables probe;
        if (!probe.build(sc)) { g_err = probe.err; delete h; return RS_ELIMIT; }
        if (!h->PT.build(sc, pick_cell_len(sc, probe.n_arr, probe.n_dep, probe.tls_maxl))) { g_err = h->PT.err; delete h; return RS_ELIMIT; }
    }
    PackedTables &PT = h->PT;
    const int C = sc->capacity;
    h->order = device_id;           // the emulation has no device: the argument carries the thread order
    h->block = block_threads > 0 ? block_threads : (C > 1024 ? 1024 : C);
    h->n_envs = n_envs;
    h->route_cont = PT.route_cont;
    h->vtype_params.assign(sc->vtype_params, sc->vtype_params + (size_t)sc->n_vtypes * VT_COLS);
    KTab &K = h->K; KCold &c = K.cold;
    K.lanes_ = PT.lanes.data(); K.links_ = PT.links.data(); K.foes_ = PT.foes.data(); K.rsteps_ = PT.rsteps.data();
    K.route_cont_ = h->route_cont.data(); K.next_link_ = PT.next_link.data(); K.notbest_ = PT.notbest.data(); K.routes_ = PT.routes.data();
    K.trip_route_ = PT.trip_route.data(); K.trip_vtype_ = PT.trip_vtype.data();
    c.trip_depart = keep_i32(h, sc->trip_depart, sc->n_trips); c.trip_next = PT.trip_next.data(); c.dep_lane = PT.dep_lane.data(); c.dep_info = PT.dep_info.data(); c.dep_first = PT.dep_first.data();
    c.vtype_params = h->vtype_params.data(); c.tls8 = PT.tls8.data(); c.fix8 = PT.fix8.data();
    c.tls_nphase = keep_i32(h, sc->tls_nphase, sc->n_signals); c.tls_ngreen = keep_i32(h, sc->tls_ngreen, sc->n_signals);
    c.tls_nlinks = keep_i32(h, sc->tls_nlinks, sc->n_signals); c.tls_state_off = PT.tls_off_p.data();
    c.tls_dur_off = keep_i32(h, sc->tls_dur_off, sc->n_signals); c.tls_yel_off = keep_i32(h, sc->tls_yel_off, sc->n_signals);
    c.tls_dur = keep_i32(h, sc->tls_dur, sc->n_tls_dur); c.tls_yellow = keep_i32(h, sc->tls_yellow, sc->n_tls_yellow);
    c.tls_init_phase = keep_i32(h, sc->tls_init_phase, sc->n_signals);
    c.fix_nphase = keep_i32(h, sc->fix_nphase, sc->n_signals); c.fix_state_off = PT.fix_off_p.data();
    c.fix_dur_off = keep_i32(h, sc->fix_dur_off, sc->n_signals); c.fix_dur = keep_i32(h, sc->fix_dur, sc->n_fix_dur);
    c.fix_init_phase = keep_i32(h, sc->fix_init_phase, sc->n_signals); c.fix_init_left = keep_i32(h, sc->fix_init_left, sc->n_signals);
    c.lane_obs = PT.lane_obs16.data(); c.obs_sig = PT.obs_sig.data(); c.sig_obs_start = keep_i32(h, sc->sig_obs_start, sc->n_signals + 1);
    c.mv_in_start = keep_i32(h, sc->mv_in_start, sc->n_signals * 12 + 1); c.mv_in_idx = keep_i32(h, sc->mv_in_idx, sc->n_mv_in);
    c.mv_out_start = keep_i32(h, sc->mv_out_start, sc->n_signals * 12 + 1); c.mv_out_idx = keep_i32(h, sc->mv_out_idx, sc->n_mv_out);
    c.pr_out_start = keep_i32(h, sc->pr_out_start, sc->n_signals + 1); c.pr_out_idx = keep_i32(h, sc->pr_out_idx, sc->n_pr_out);
    c.trips_cum = keep_i32(h, sc->trips_cum, sc->horizon + 2);
    K.maxlen = PT.maxlen; K.occ_unit = PT.occ_unit; K.n_trips = sc->n_trips; K.tls_maxl = PT.tls_maxl; K.kmax = sc->kmax;
    K.n_lanes = sc->n_lanes; K.n_cells = PT.n_cells; K.n_signals = sc->n_signals; K.n_obs = sc->n_obs; K.n_vtypes = sc->n_vtypes;
    h->ratio = p->step_ratio > 1 ? p->step_ratio : 1;
    K.horizon = sc->horizon; K.capacity = C; K.step_length = sc->step_length; K.yellow_length = sc->yellow_length * h->ratio; K.lmax = PT.lmax;
    K.n_arr = PT.n_arr; K.n_dep = PT.n_dep;
    h->P.seed = p->seed; h->P.env_base = env_base; h->P.max_distance = p->max_distance; h->P.sigma = p->sigma;
    h->P.speed_dev = p->speed_dev; h->P.fixed_program = p->fixed_program; h->P.tls_expiry = p->tls_hold == 0; h->P.n_envs = n_envs;
    const size_t N = (size_t)n_envs, NC = N * C, S = (size_t)sc->n_signals;
    h->G.nc = NC; h->slab.assign(State::bytes(NC), 0); h->G.base = h->slab.data();
    h->O.n = n_envs; h->O.o = sc->n_obs; h->O.s = sc->n_signals; h->O.lm = PT.lmax;
    h->outb.assign(h->O.bytes(), 0); h->O.base = h->outb.data();
    h->env.assign(N * 4, 0); h->tls.assign(N * S * TLS_W, 0); h->stats.assign(N * ST_N, 0); h->actions.assign(N * S, 0);
    h->dep_next.assign(N * K.n_dep, 0);
    h->mail.assign(N * (size_t)((C + 31) / 32), 0u);
    h->G.env = h->env.data(); h->G.tls = h->tls.data(); h->G.stats = h->stats.data(); h->G.dep_next = h->dep_next.data(); h->G.mail = h->mail.data();
    h->G.trip_log = nullptr;
    if (p->trip_log) { h->trip_log.assign(N * (size_t)sc->n_trips * 4, 0); h->G.trip_log = h->trip_log.data(); }
    h->lds = lds_carve(&h->L, C, K.n_cells, K.n_arr, K.n_dep, sc->n_obs, sc->n_signals, sc->n_vtypes, K.tls_maxl);
    if (!lds_fix_matches(h->L, C)) { g_err = "the layout of the working memory does not match the kernel's literals (lds_carve / LdsFix)"; delete h; return RS_EINVAL; }
    h->L.cell_inv = h->PT.cell_inv;
    h->smem.assign(h->lds + 64, 0);
    State &G = h->G; Out &O = h->O;
    const int64_t n = n_envs, cc = C, s = sc->n_signals, o = sc->n_obs, lmax = PT.lmax;
    set_buf(h, RS_BUF_LANE_AGG, O.lane_agg(), RS_F32, 3, n, o, 5); set_buf(h, RS_BUF_DRQ_NORM, O.drq_norm(), RS_F32, 3, n, o, 5);
    set_buf(h, RS_BUF_PHASE, O.phase(), RS_I32, 2, n, s); set_buf(h, RS_BUF_MPLIGHT, O.mplight(), RS_I32, 3, n, s, 13);
    set_buf(h, RS_BUF_WAVE, O.wave(), RS_I32, 3, n, s, 12); set_buf(h, RS_BUF_WAIT, O.wait(), RS_F32, 2, n, s);
    set_buf(h, RS_BUF_WAIT_NORM, O.wait_norm(), RS_F32, 2, n, s); set_buf(h, RS_BUF_PRESSURE, O.pressure(), RS_I32, 2, n, s);
    set_buf(h, RS_BUF_QUEUE_SUM, O.queue_sum(), RS_I32, 2, n, s); set_buf(h, RS_BUF_QUEUE_MAX, O.queue_max(), RS_I32, 2, n, s);
    set_buf(h, RS_BUF_ACTIONS, h->actions.data(), RS_I32, 2, n, s); set_buf(h, RS_BUF_ENV, G.env, RS_I32, 2, n, 4);
    set_buf(h, RS_BUF_TLS, G.tls, RS_I32, 3, n, s, TLS_W);
    set_buf(h, RS_BUF_VEH_POS, G.pos(), RS_F32, 2, n, cc); set_buf(h, RS_BUF_VEH_SPEED, G.speed(), RS_F32, 2, n, cc);
    set_buf(h, RS_BUF_VEH_ACCEL, G.accel(), RS_F32, 2, n, cc); set_buf(h, RS_BUF_VEH_TLOSS, G.tloss(), RS_F32, 2, n, cc);
    set_buf(h, RS_BUF_VEH_LANE, G.lane(), RS_U16, 2, n, cc); set_buf(h, RS_BUF_VEH_TRIP, G.trip(), RS_U16, 2, n, cc);
    set_buf(h, RS_BUF_VEH_CURSOR, G.cursor(), RS_U16, 2, n, cc); set_buf(h, RS_BUF_VEH_SWAIT, G.swait(), RS_U16, 2, n, cc);
    set_buf(h, RS_BUF_VEH_RWAIT, G.rwait(), RS_U16, 2, n, cc); set_buf(h, RS_BUF_VEH_DEPART, G.depart(), RS_U16, 2, n, cc);
    set_buf(h, RS_BUF_VEH_OWNER, G.owner(), RS_U8, 2, n, cc); set_buf(h, RS_BUF_STATS, G.stats, RS_I64, 2, n, ST_N);
    set_buf(h, RS_BUF_DRQ_NORM_F16, O.drq_f16(), RS_F16, 4, n, s, lmax, 5); set_buf(h, RS_BUF_VEH_SF, G.sf(), RS_F32, 2, n, cc);
    set_buf(h, RS_BUF_VEH_WTOT, G.wtot(), RS_U16, 2, n, cc);
    set_buf(h, RS_BUF_TRIP_LOG, G.trip_log, RS_I32, 3, n, p->trip_log ? sc->n_trips : 0, 4);
    set_buf(h, RS_BUF_DEP_NEXT, G.dep_next, RS_U16, 2, n, K.n_dep);
    set_buf(h, RS_BUF_VEH_COOP, G.coop(0), RS_U32, 2, n, cc); set_buf(h, RS_BUF_VEH_COOPLEAD, G.cooplead(0), RS_U32, 2, n, cc);
    set_buf(h, RS_BUF_ARRIVALS, O.arrivals(), RS_I32, 2, n, s); set_buf(h, RS_BUF_DEPARTURES, O.departures(), RS_I32, 2, n, s);
    set_buf(h, RS_BUF_MPLIGHT_FULL, O.mplight_full(), RS_F32, 3, n, s, 49);
    set_buf(h, RS_BUF_LANE_ARRIVALS, O.lane_arr(), RS_I32, 2, n, sc->n_obs);
    set_buf(h, RS_BUF_VEH_COOP_ODD, G.coop(1), RS_U32, 2, n, cc); set_buf(h, RS_BUF_VEH_COOPLEAD_ODD, G.cooplead(1), RS_U32, 2, n, cc);
    set_buf(h, RS_BUF_VEH_MAIL, G.mail, RS_U32, 2, n, (cc + 31) / 32);
    *out = h;
    return rs_reset(h, nullptr);
}
void rs_destroy(rs_handle h) { delete h; }
const char *rs_last_error(rs_handle h) { return h ? h->err.c_str() : g_err.c_str(); }
int rs_reset(rs_handle h, void *) {
    const KTab &T = h->K; const State &G = h->G;
    const int C = T.capacity, S = T.n_signals;
    for (int env = 0; env < h->n_envs; ++env) {
        const size_t eo = (size_t)env * C;
        for (int s = 0; s < C; ++s) {
            G.lane()[eo + s] = LANE_NONE; G.trip()[eo + s] = TRIP_NONE; G.owner()[eo + s] = OWNER_NONE;
            G.rwait()[eo + s] = 0; G.swait()[eo + s] = 0; G.cursor()[eo + s] = 0; G.depart()[eo + s] = 0; G.wtot()[eo + s] = 0;
            G.pos()[eo + s] = 0.0f; G.speed()[eo + s] = 0.0f; G.accel()[eo + s] = 0.0f; G.tloss()[eo + s] = 0.0f; G.sf()[eo + s] = 1.0f;
            G.coop(0)[eo + s] = COOP_NONE; G.coop(1)[eo + s] = COOP_NONE; G.cooplead(0)[eo + s] = COOP_NONE; G.cooplead(1)[eo + s] = COOP_NONE;
        }
        for (int s = 0; s < S; ++s) {
            int ph, left;
            if (h->P.fixed_program) { ph = T.cold.fix_init_phase[s]; left = T.cold.fix_init_left[s]; }
            else { ph = T.cold.tls_init_phase[s]; left = T.cold.tls_dur[T.cold.tls_dur_off[s] + ph]; }
            G.tls[(env * S + s) * TLS_W + 0] = ph; G.tls[(env * S + s) * TLS_W + 1] = left; G.tls[(env * S + s) * TLS_W + 2] = 0; G.tls[(env * S + s) * TLS_W + 3] = 0;
        }
        for (int d = 0; d < T.n_dep; ++d) G.dep_next[(size_t)env * T.n_dep + d] = T.cold.dep_first[d];
        for (int i = 0; i < (C + 31) / 32; ++i) G.mail[(size_t)env * ((C + 31) / 32) + i] = 0u;
        for (int i = 0; i < 4; ++i) G.env[env * 4 + i] = 0;
        for (int i = 0; i < ST_N; ++i) G.stats[(size_t)env * ST_N + i] = 0;
        if (G.trip_log) for (int i = 0; i < T.n_trips * 4; ++i) G.trip_log[(size_t)env * T.n_trips * 4 + i] = 0;
    }
    run_step(h, 0, 0);
    return RS_OK;
}
int rs_step(rs_handle h, const int32_t *actions, int32_t, void *) {
    if (actions) memcpy(h->actions.data(), actions, h->actions.size() * 4);
    run_step(h, h->K.step_length * h->ratio, 1);
    return RS_OK;
}
int rs_ticks(rs_handle h, int32_t n, void *) { run_step(h, n, 0); return RS_OK; }
int rs_step_sim(rs_handle h, int32_t n, void *) { run_step(h, n, 0, 0); return RS_OK; }
int rs_set_outputs(rs_handle h, uint64_t buffer_mask) {
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
int rs_sync(rs_handle) { return RS_OK; }
int rs_reinit_signals(rs_handle h, void *) {
    const KTab &T = h->K; const State &G = h->G;
    const int C = T.capacity, S = T.n_signals;
    for (int env = 0; env < h->n_envs; ++env) {
        const size_t eo = (size_t)env * C;
        for (int s = 0; s < C; ++s) { G.owner()[eo + s] = OWNER_NONE; G.rwait()[eo + s] = 0; }
        for (int s = 0; s < S; ++s) {
            if (!h->P.fixed_program) G.tls[(env * S + s) * TLS_W + 1] = T.cold.tls_dur[T.cold.tls_dur_off[s] + G.tls[(env * S + s) * TLS_W + 0]];
            G.tls[(env * S + s) * TLS_W + 2] = 0; G.tls[(env * S + s) * TLS_W + 3] = 0;
        }
    }
    run_step(h, 0, 0);
    return RS_OK;
}
int rs_act_random(rs_handle h, uint32_t step_key, void *) {
    const int S = h->K.n_signals;
    for (int i = 0; i < h->n_envs * S; ++i) {
        const int env = i / S, s = i - env * S;
        const uint32_t hh = d_hash(h->P.seed ^ 0xA5A5A5A5u, (uint32_t)(h->P.env_base + env), (uint32_t)s, step_key, 7u);
        h->actions[i] = (int32_t)(hh % (uint32_t)h->K.cold.tls_ngreen[s]);
    }
    return RS_OK;
}
int rs_act_maxwave(rs_handle h, const int32_t *phase_pairs, int32_t n_pairs, const int32_t *valid, const int32_t *order, int32_t use_pressure, void *) {
    const int S = h->K.n_signals;
    if (!(h->out_mask & (use_pressure ? OUT_MPLIGHT : OUT_WAVE))) return RS_EINVAL;     // (as the HIP library: the rows it would read are switched off)
    if (phase_pairs) {
        h->pairs.assign(phase_pairs, phase_pairs + n_pairs * 2); h->valid.assign(valid, valid + S * n_pairs); h->ordr.assign(order, order + S * n_pairs);
        h->n_pairs = n_pairs;
    }
    for (int i = 0; i < h->n_envs * S; ++i) {
        const int s = i % S;
        const int32_t *obs = use_pressure ? h->O.mplight() + (size_t)i * 13 + 1 : h->O.wave() + (size_t)i * 12;
        bool have = false; int best = 0, best_act = 0;
        for (int j = 0; j < h->n_pairs; ++j) {
            const int p = h->ordr[s * h->n_pairs + j];
            if (p < 0) break;
            const int act = h->valid[s * h->n_pairs + p];
            if (act < 0) continue;
            const int press = obs[h->pairs[p * 2]] + obs[h->pairs[p * 2 + 1]];
            if (!have || press > best) { have = true; best = press; best_act = act; }
        }
        h->actions[i] = best_act;
    }
    return RS_OK;
}
// the static agents + the step of a group of handles in one call (include/resco_sim.h); the policy network is device code only
int rs_group_step(const rs_handle *hs, int32_t n, const rs_group_agent *agent, int32_t n_steps) {
    if (!hs || n <= 0 || n_steps <= 0) return RS_EINVAL;
    const int kind = agent ? agent->kind : RS_AGENT_NONE;
    if (kind == RS_AGENT_IDQN || kind < 0 || kind > RS_AGENT_IDQN) return RS_EINVAL;
    for (int i = 0; i < n; ++i)
        if ((kind == RS_AGENT_MAXWAVE || kind == RS_AGENT_MAXPRESSURE) && hs[i]->n_pairs == 0) return RS_EINVAL;
    for (int k = 0; k < n_steps; ++k)
        for (int i = 0; i < n; ++i) {
            int rc = RS_OK;
            if (kind == RS_AGENT_RANDOM) rc = rs_act_random(hs[i], agent->step_key + (uint32_t)k, nullptr);
            else if (kind == RS_AGENT_MAXWAVE || kind == RS_AGENT_MAXPRESSURE) rc = rs_act_maxwave(hs[i], nullptr, hs[i]->n_pairs, nullptr, nullptr, kind == RS_AGENT_MAXPRESSURE, nullptr);
            if (rc != RS_OK) return rc;
            rs_step(hs[i], nullptr, 1, nullptr);
        }
    return RS_OK;
}
int rs_get_buffer(rs_handle h, int32_t which, void **dev_ptr, int64_t shape[4], int32_t *ndim, int32_t *dtype) {
    if (!h || which < 0 || which >= RS_BUF_COUNT) return RS_EINVAL;
    auto &B = h->bufs[which];
    if (dev_ptr) *dev_ptr = B.ptr;
    if (shape) for (int i = 0; i < 4; ++i) shape[i] = B.shape[i];
    if (ndim) *ndim = B.ndim;
    if (dtype) *dtype = B.dtype;
    return RS_OK;
}
int rs_read_buffer(rs_handle h, int32_t which, void *host_dst, int64_t nbytes) {
    auto &B = h->bufs[which];
    if ((size_t)nbytes != B.bytes) return RS_EINVAL;
    if (B.bytes) memcpy(host_dst, B.ptr, B.bytes);
    return RS_OK;
}
int rs_stats(rs_handle h, int64_t *out) { return rs_read_buffer(h, RS_BUF_STATS, out, (int64_t)h->n_envs * ST_N * 8); }
int rs_set_seed(rs_handle h, uint32_t seed) { h->P.seed = seed; return RS_OK; }
// snapshots: every buffer, in id order
struct Snap { std::vector<std::vector<char>> b; };
int rs_snapshot(rs_handle h, void **snap) {
    Snap *S = new Snap();
    for (int i = 0; i < RS_BUF_COUNT; ++i) S->b.emplace_back((char *)h->bufs[i].ptr, (char *)h->bufs[i].ptr + (h->bufs[i].ptr ? h->bufs[i].bytes : 0));
    *snap = S;
    return RS_OK;
}
int rs_restore(rs_handle h, const void *snap) {
    const Snap *S = (const Snap *)snap;
    for (int i = 0; i < RS_BUF_COUNT; ++i) if (!S->b[i].empty()) memcpy(h->bufs[i].ptr, S->b[i].data(), S->b[i].size());
    return RS_OK;
}
void rs_snapshot_free(rs_handle, void *snap) { delete (Snap *)snap; }
int rs_timing(rs_handle, int32_t) { return RS_OK; }
int rs_timing_read(rs_handle, float *ms, int32_t *n) { if (ms) *ms = 0.0f; if (n) *n = 0; return RS_OK; }
int rs_phase_profile(rs_handle, int32_t, uint64_t *out) { if (out) memset(out, 0, 16 * 8); return RS_OK; }
int rs_info(rs_handle h, int32_t *n_envs, int32_t *block_threads, int32_t *lds_bytes, int32_t *max_lanes) {
    if (n_envs) *n_envs = h->n_envs; if (block_threads) *block_threads = h->block; if (lds_bytes) *lds_bytes = (int32_t)h->lds; if (max_lanes) *max_lanes = h->K.lmax;
    return RS_OK;
}
}
