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

#define LANE_NONE 0xFFFFu
#define LANE_PENDING 0xFFFEu
#define OWNER_NONE 0xFFu
#define NIL 0x7FFF              /* list terminator (15-bit slot ids; bit 15 of a head = lane has a moving vehicle) */
#define HALT_SPEED 0.1f
#define STOP_OFFSET 1.0f
#define ARR_NONE 65535
#define FOE_GAP_Q 40
#define MAX_HOPS 6
#define BIGF 1.0e30f
#define SG_ADVANTAGE 10.0f
#define URGENT_DIST 50.0f
#define SWAP_WAIT 20
#define SWAP_EVERY 4

enum { VT_LENGTH, VT_MINGAP, VT_ACCEL, VT_DECEL, VT_TAU, VT_SIGMA, VT_MAXSPEED, VT_SF_MEAN, VT_SF_DEV, VT_EMERGENCY, VT_COLS };
enum { TLS_R = 0, TLS_Y = 1, TLS_g = 2, TLS_G = 3 };
enum { ST_INSERTED, ST_ARRIVED, ST_DURATION, ST_DEPDELAY, ST_WAITING, ST_TLOSS, ST_ACTIVE, ST_PENDING, ST_ACTIVE_TICKS, ST_TICKS, ST_N };

#include "resco_kernels.h"
#include "resco_policy.h"

// ------------------------------------------------------------------------------------------------ host side
struct rs_sim {
    int device = 0;
    int n_envs = 0, env_base = 0, block = 256;
    size_t lds = 0;
    Tab T{};
    KTab K{};
    State G{};
    Out O{};
    KParams P{};
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

static const size_t kDtypeSize[] = {4, 4, 2, 1, 2, 8};
static void set_buf(rs_sim *h, int which, void *ptr, int dtype, int ndim, int64_t a, int64_t b = 1, int64_t c = 1, int64_t d = 1) {
    auto &B = h->bufs[which];
    B.ptr = ptr; B.dtype = dtype; B.ndim = ndim;
    B.shape[0] = a; B.shape[1] = b; B.shape[2] = c; B.shape[3] = d;
    B.bytes = (size_t)(a * b * c * d) * kDtypeSize[dtype];
}

// the step kernel is instantiated for the capacities compile_scenario produces (constant LDS offsets); any other
// capacity runs the generic instantiation
typedef void (*step_kernel_fn)(KTab, State, Out, KParams, const int32_t *);
static const int kStepCaps[] = {0, 128, 256, 512, 1024};
static step_kernel_fn step_kernel_for(int capacity) {
    switch (capacity) {
        case 128: return rs_step_kernel<128>;
        case 256: return rs_step_kernel<256>;
        case 512: return rs_step_kernel<512>;
        case 1024: return rs_step_kernel<1024>;
        default: return rs_step_kernel<0>;
    }
}

// every synchronous entry point waits for the handle's own stream AND the caller stream of the last launch
static hipError_t wait_idle(rs_sim *h) {
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess && h->last && h->last != h->stream) e = hipStreamSynchronize(h->last);
    return e;
}

extern "C" int rs_create(const rs_scenario *sc, const rs_params *p, int32_t n_envs, int32_t env_base, int32_t device_id,
                         int32_t block_threads, rs_handle *out) {
    if (!sc || !p || !out || n_envs <= 0) { g_create_err = "rs_create: bad argument"; return RS_EINVAL; }
    rs_sim *h = new (std::nothrow) rs_sim();
    if (!h) return RS_ENOMEM;
    auto fail = [&](int rc) { g_create_err = h->err; rs_destroy(h); return rc; };
    h->device = device_id; h->n_envs = n_envs; h->env_base = env_base;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { h->err = "no HIP device visible (this library has no CPU fallback)"; return fail(RS_EHIP); }
    if (hipSetDevice(device_id) != hipSuccess) { h->err = "hipSetDevice failed"; return fail(RS_EHIP); }
    if (sc->step_length <= 0 || sc->yellow_length < 0 || sc->yellow_length >= sc->step_length) {
        h->err = "need 0 <= yellow_length < step_length"; return fail(RS_EINVAL);
    }
    const int C = sc->capacity;
    if (C < 64 || (C & (C - 1)) || C > 16384) { h->err = "capacity must be a power of two in [64, 16384]"; return fail(RS_ELIMIT); }
    if (sc->n_lanes >= 0xFFFE || sc->n_trips >= 0xFFFF || sc->n_routes > 0xFFFF || sc->n_vtypes > 255 || sc->n_signals > 254) {
        h->err = "scenario exceeds id widths (lanes/trips/routes u16, vtypes/signals u8)"; return fail(RS_ELIMIT);
    }
    Tab &T = h->T;
    int rc;
#define X(name, type, count)                                                                  \
    if ((rc = dev_upload<type>(h, &T.name, sc->name, (size_t)(sc->count)))) return fail(rc);
    RS_TABLES(X)
#undef X
    T.n_lanes = sc->n_lanes; T.n_links = sc->n_links; T.n_edges = sc->n_edges; T.n_routes = sc->n_routes;
    T.n_trips = sc->n_trips; T.n_signals = sc->n_signals; T.n_obs = sc->n_obs; T.n_vtypes = sc->n_vtypes;
    T.horizon = sc->horizon; T.capacity = C; T.step_length = sc->step_length; T.yellow_length = sc->yellow_length;
    std::vector<int32_t> obs_sig((size_t)sc->n_obs > 0 ? sc->n_obs : 1, 0);
    int lmax = 1;
    for (int s = 0; s < sc->n_signals; ++s) {
        for (int oi = sc->sig_obs_start[s]; oi < sc->sig_obs_start[s + 1]; ++oi) obs_sig[oi] = s;
        int n = sc->sig_obs_start[s + 1] - sc->sig_obs_start[s];
        if (n > lmax) lmax = n;
    }
    T.lmax = lmax;
    if ((rc = dev_upload<int32_t>(h, &T.obs_sig, obs_sig.data(), obs_sig.size()))) return fail(rc);
    h->tls_ngreen.assign(sc->tls_ngreen, sc->tls_ngreen + sc->n_signals);
    int tls_maxl = 1;
    for (int s = 0; s < sc->n_signals; ++s) if (sc->tls_nlinks[s] > tls_maxl) tls_maxl = sc->tls_nlinks[s];
    // ---- packed 16-byte records for the step kernel
    {
        if (sc->n_route_steps >= 0xFFFF || sc->n_foes >= 0xFFFF || sc->n_links >= 0x7FFF || sc->n_edges >= 0xFFFF || sc->n_obs >= 0x7FFF) {
            h->err = "scenario exceeds packed-table id widths (route steps / foes / links / edges u16)"; return fail(RS_ELIMIT);
        }
        std::vector<int16_t> link_arr((size_t)sc->n_links, -1);
        int n_foe_targets = 0;
        for (int l = 0; l < sc->n_links; ++l)
            for (int i = sc->link_foe_start[l]; i < sc->link_foe_start[l] + sc->link_foe_cnt[l]; ++i) {
                int f = sc->foe_link[i];
                if (link_arr[f] < 0) link_arr[f] = (int16_t)n_foe_targets++;
            }
        std::vector<LaneRec> lanes((size_t)sc->n_lanes);
        for (int l = 0; l < sc->n_lanes; ++l) {
            LaneRec &R = lanes[l];
            R.len = sc->lane_len[l]; R.vmax = sc->lane_vmax[l];
            R.link_start = (uint16_t)sc->lane_link_start[l];
            if (sc->lane_link_cnt[l] > 255) { h->err = "more than 255 links on one lane"; return fail(RS_ELIMIT); }
            R.link_cnt = (uint8_t)sc->lane_link_cnt[l];
            int e = sc->lane_edge[l];
            int nl = e >= 0 ? sc->edge_nlanes[e] : 0;
            R.flags = (uint8_t)((sc->lane_internal[l] ? LF_INTERNAL : 0u) | ((unsigned)nl << 2));
            R.cell0 = 0;        // filled below
            R.edge_lane0 = (uint16_t)(e >= 0 ? sc->edge_lane0[e] : 0);
        }
        // list cells: floor(len / CELL_LEN) + 1 per lane, lanes in index order (lanes of one edge are consecutive
        // and equally long, so their cell blocks are consecutive and equally sized -- relied on by the lane change)
        int n_cells = 0;
        std::vector<int> lane_nc((size_t)sc->n_lanes);
        for (int l = 0; l < sc->n_lanes; ++l) {
            lane_nc[l] = (int)(sc->lane_len[l] * CELL_INV) + 1;
            lanes[l].cell0 = (uint16_t)n_cells;
            n_cells += lane_nc[l];
        }
        if (n_cells >= 0x7FFF) { h->err = "too many list cells"; return fail(RS_ELIMIT); }
        for (int e = 0; e < sc->n_edges; ++e)
            for (int j = 1; j < sc->edge_nlanes[e]; ++j)
                if (lane_nc[sc->edge_lane0[e] + j] != lane_nc[sc->edge_lane0[e]]) { h->err = "lanes of one edge differ in length"; return fail(RS_ELIMIT); }
        {
            float max_len = 0.0f, max_gap = 0.0f;
            for (int v = 0; v < sc->n_vtypes; ++v) {
                if (sc->vtype_params[v * VT_COLS + VT_LENGTH] > max_len) max_len = sc->vtype_params[v * VT_COLS + VT_LENGTH];
                if (sc->vtype_params[v * VT_COLS + VT_MINGAP] > max_gap) max_gap = sc->vtype_params[v * VT_COLS + VT_MINGAP];
            }
            // the insertion check scans cell 0 only: everything that can be in the way must sit there
            if (2.0f * max_len + max_gap >= CELL_LEN) { h->err = "vehicle length + minGap + length must stay below the list cell size"; return fail(RS_ELIMIT); }
        }
        std::vector<int16_t> lane_obs16((size_t)sc->n_lanes);
        for (int l = 0; l < sc->n_lanes; ++l) lane_obs16[l] = (int16_t)sc->lane_obs[l];
        std::vector<LinkRec> links((size_t)sc->n_links);
        for (int l = 0; l < sc->n_links; ++l) {
            LinkRec &R = links[l];
            R.to_lane = (uint16_t)sc->link_to_lane[l]; R.to_edge = (uint16_t)sc->link_to_edge[l];
            R.foe_start = (uint16_t)sc->link_foe_start[l];
            R.via2 = sc->link_via2[l] >= 0 ? (uint16_t)sc->link_via2[l] : (uint16_t)0xFFFF;
            R.arr_idx = link_arr[l];
            R.tls = sc->link_tls[l] >= 0 ? (uint8_t)sc->link_tls[l] : (uint8_t)0xFF;
            R.tls_pos = sc->link_tls[l] >= 0 ? (uint8_t)sc->link_tls_pos[l] : (uint8_t)0;
            if (sc->link_foe_cnt[l] > 255 || (sc->link_tls[l] >= 0 && sc->link_tls_pos[l] > 255)) { h->err = "foe count / TLS link index exceeds u8"; return fail(RS_ELIMIT); }
            R.foe_cnt = (uint8_t)sc->link_foe_cnt[l];
            R.flags = (uint8_t)((sc->link_minor[l] ? KF_MINOR : 0u) | (sc->link_cont[l] ? KF_CONT : 0u) | (sc->link_via1[l] >= 0 ? KF_VIA1 : 0u));
            R.dest_k = (uint8_t)(sc->link_dest_lane[l] - sc->edge_lane0[sc->link_to_edge[l]]);
            R.pad = 0;
            R.dest = lanes[sc->link_to_lane[l]];
        }
        std::vector<FoeRec> foes((size_t)(sc->n_foes > 0 ? sc->n_foes : 1));
        for (int i = 0; i < sc->n_foes; ++i) {
            int f = sc->foe_link[i];
            FoeRec &R = foes[i];
            R.arr_idx = link_arr[f];
            R.tls = sc->link_tls[f] >= 0 ? (uint8_t)sc->link_tls[f] : (uint8_t)0xFF;
            R.tls_pos = sc->link_tls[f] >= 0 ? (uint8_t)sc->link_tls_pos[f] : (uint8_t)0;
            const int v1 = sc->link_via1[f], v2 = sc->link_via2[f];
            R.via1_cell0 = v1 >= 0 ? lanes[v1].cell0 : (uint16_t)0xFFFF;
            R.via2_cell0 = v2 >= 0 ? lanes[v2].cell0 : (uint16_t)0xFFFF;
            R.via1_nc = v1 >= 0 ? (uint8_t)lane_nc[v1] : (uint8_t)0;
            R.via2_nc = v2 >= 0 ? (uint8_t)lane_nc[v2] : (uint8_t)0;
            memset(R.pad, 0, sizeof(R.pad));
        }
        std::vector<RStep> rsteps((size_t)(sc->n_route_steps > 0 ? sc->n_route_steps : 1));
        std::vector<RouteRec> routes((size_t)sc->n_routes);
        std::vector<int16_t> lane_dep((size_t)sc->n_lanes, -1);
        int n_dep = 0;
        for (int r = 0; r < sc->n_routes; ++r) {
            const int rs = sc->route_start[r], re = sc->route_start[r + 1];
            for (int q = rs; q < re; ++q) {
                RStep &R = rsteps[q];
                R.edge = (uint16_t)sc->route_edge[q];
                const bool last = q + 1 >= re;
                R.next_edge = last ? (uint16_t)0xFFFF : (uint16_t)sc->route_edge[q + 1];
                R.next_mask2 = last ? 0u : sc->route_mask2[q + 1];
                R.next_mask1 = last ? 0u : sc->route_mask1[q + 1];
                R.tlsdist = sc->route_tlsdist[q];
            }
            const uint32_t m = sc->route_mask2[rs];
            const int e = sc->route_edge[rs];
            int k = 0;
            while (k < 31 && !((m >> k) & 1u)) k += 1;
            if (k >= sc->edge_nlanes[e]) k = 0;
            const int dl = sc->edge_lane0[e] + k;
            if (lane_dep[dl] < 0) lane_dep[dl] = (int16_t)n_dep++;
            routes[r].start = (uint32_t)rs; routes[r].depart_lane = (uint16_t)dl; routes[r].depart_arr = lane_dep[dl];
            routes[r].depart_cell0 = lanes[dl].cell0; routes[r].depart_len = sc->lane_len[dl];
            {   // choose_link(departure lane, first route step) + the approach-register flag, as cache_link() computes it
                int link = -1;
                if (re - rs >= 2) {
                    const int ne = sc->route_edge[rs + 1];
                    const uint32_t pref = sc->route_mask2[rs + 1], okm = sc->route_mask1[rs + 1];
                    int best = -1, any = -1;
                    for (int l = sc->lane_link_start[dl]; l < sc->lane_link_start[dl] + sc->lane_link_cnt[dl]; ++l) {
                        if (sc->link_to_edge[l] != ne) continue;
                        const int kk = sc->link_dest_lane[l] - sc->edge_lane0[ne];
                        if ((pref >> kk) & 1u) { link = l; break; }
                        if (best < 0 && ((okm >> kk) & 1u)) best = l;
                        if (any < 0) any = l;
                    }
                    if (link < 0) link = best >= 0 ? best : any;
                }
                routes[r].first_link = link < 0 ? (uint16_t)NLINK_NONE : (uint16_t)(link | (link_arr[link] >= 0 ? NLINK_ARR : 0));
            }
        }
        std::vector<uint16_t> trip_route((size_t)sc->n_trips);
        std::vector<uint8_t> trip_vtype((size_t)sc->n_trips);
        for (int k = 0; k < sc->n_trips; ++k) { trip_route[k] = (uint16_t)sc->trip_route[k]; trip_vtype[k] = (uint8_t)sc->trip_vtype[k]; }
        // choose_link() of every (route step, lane of that step's edge): the look-ahead and the lane hand-over read it
        int kmax = 1;
        for (int e = 0; e < sc->n_edges; ++e) if (sc->edge_nlanes[e] > kmax) kmax = sc->edge_nlanes[e];
        std::vector<uint16_t> next_link((size_t)(sc->n_route_steps > 0 ? sc->n_route_steps : 1) * kmax, (uint16_t)0xFFFF);
        for (int r = 0; r < sc->n_routes; ++r) {
            const int rs = sc->route_start[r], re = sc->route_start[r + 1];
            for (int q = rs; q + 1 < re; ++q) {
                const int e = sc->route_edge[q], ne = sc->route_edge[q + 1];
                const uint32_t pref = sc->route_mask2[q + 1], okm = sc->route_mask1[q + 1];
                for (int k = 0; k < sc->edge_nlanes[e]; ++k) {
                    const int ln = sc->edge_lane0[e] + k;
                    int link = -1, best = -1, any = -1;
                    for (int l = sc->lane_link_start[ln]; l < sc->lane_link_start[ln] + sc->lane_link_cnt[ln]; ++l) {
                        if (sc->link_to_edge[l] != ne) continue;
                        const int kk = sc->link_dest_lane[l] - sc->edge_lane0[ne];
                        if ((pref >> kk) & 1u) { link = l; break; }
                        if (best < 0 && ((okm >> kk) & 1u)) best = l;
                        if (any < 0) any = l;
                    }
                    if (link < 0) link = best >= 0 ? best : any;
                    if (link >= 0) next_link[(size_t)q * kmax + k] = (uint16_t)link;
                }
            }
        }
        std::vector<uint8_t> tls8((size_t)(sc->n_tls_states > 0 ? sc->n_tls_states : 1)), fix8((size_t)(sc->n_fix_states > 0 ? sc->n_fix_states : 1));
        for (int i = 0; i < sc->n_tls_states; ++i) tls8[i] = (uint8_t)sc->tls_states[i];
        for (int i = 0; i < sc->n_fix_states; ++i) fix8[i] = (uint8_t)sc->fix_states[i];
        KTab &K = h->K;
        const int16_t *lane_obs_dev = nullptr;
        const uint8_t *tls8_dev = nullptr, *fix8_dev = nullptr;
        if ((rc = dev_upload<LaneRec>(h, &K.lanes, lanes.data(), lanes.size())) || (rc = dev_upload<LinkRec>(h, &K.links, links.data(), links.size())) ||
            (rc = dev_upload<FoeRec>(h, &K.foes, foes.data(), foes.size())) || (rc = dev_upload<RStep>(h, &K.rsteps, rsteps.data(), rsteps.size())) ||
            (rc = dev_upload<RouteRec>(h, &K.routes, routes.data(), routes.size())) ||
            (rc = dev_upload<uint16_t>(h, &K.next_link, next_link.data(), next_link.size())) ||
            (rc = dev_upload<uint16_t>(h, &K.trip_route, trip_route.data(), trip_route.size())) ||
            (rc = dev_upload<uint8_t>(h, &K.trip_vtype, trip_vtype.data(), trip_vtype.size())) ||
            (rc = dev_upload<int16_t>(h, &lane_obs_dev, lane_obs16.data(), lane_obs16.size())) ||
            (rc = dev_upload<uint8_t>(h, &tls8_dev, tls8.data(), tls8.size())) || (rc = dev_upload<uint8_t>(h, &fix8_dev, fix8.data(), fix8.size())))
            return fail(rc);
        K.route_mask2 = T.route_mask2;
        K.kmax = kmax;
        KCold cold{};
        cold.trip_depart = T.trip_depart; cold.trips_cum = T.trips_cum; cold.vtype_params = T.vtype_params;
        cold.tls8 = tls8_dev; cold.fix8 = fix8_dev; cold.lane_obs = lane_obs_dev;
        cold.tls_nphase = T.tls_nphase; cold.tls_ngreen = T.tls_ngreen; cold.tls_nlinks = T.tls_nlinks; cold.tls_state_off = T.tls_state_off;
        cold.tls_dur_off = T.tls_dur_off; cold.tls_yel_off = T.tls_yel_off; cold.tls_dur = T.tls_dur; cold.tls_yellow = T.tls_yellow;
        cold.fix_nphase = T.fix_nphase; cold.fix_state_off = T.fix_state_off; cold.fix_dur_off = T.fix_dur_off; cold.fix_dur = T.fix_dur;
        cold.obs_sig = T.obs_sig; cold.sig_obs_start = T.sig_obs_start; cold.mv_in_start = T.mv_in_start; cold.mv_in_idx = T.mv_in_idx;
        cold.mv_out_start = T.mv_out_start; cold.mv_out_idx = T.mv_out_idx; cold.pr_out_start = T.pr_out_start; cold.pr_out_idx = T.pr_out_idx;
        if ((rc = dev_upload<KCold>(h, &K.cold, &cold, 1))) return fail(rc);
        K.n_trips = sc->n_trips; K.tls_maxl = tls_maxl;
        K.n_lanes = sc->n_lanes; K.n_cells = n_cells; K.n_signals = sc->n_signals; K.n_obs = sc->n_obs; K.n_vtypes = sc->n_vtypes; K.horizon = sc->horizon;
        K.capacity = C; K.step_length = sc->step_length; K.yellow_length = sc->yellow_length; K.lmax = lmax;
        K.n_arr = n_foe_targets > 0 ? n_foe_targets : 1;
        K.n_dep = n_dep > 0 ? n_dep : 1;
        T.n_arr = K.n_arr;
    }


    h->P.seed = p->seed; h->P.env_base = env_base; h->P.max_distance = p->max_distance; h->P.sigma = p->sigma;
    h->P.speed_dev = p->speed_dev; h->P.fixed_program = p->fixed_program; h->P.n_envs = n_envs;

    const size_t N = (size_t)n_envs, NC = N * C, S = (size_t)sc->n_signals, NO = (size_t)sc->n_obs;
    State &G = h->G;
    Out &O = h->O;
    {
        char *slab = nullptr, *outb = nullptr;
        G.nc = NC;
        O.n = n_envs; O.o = sc->n_obs; O.s = sc->n_signals; O.lm = lmax;
        if ((rc = dev_alloc(h, &slab, State::bytes(NC))) || (rc = dev_alloc(h, &outb, O.bytes())) ||
            (rc = dev_alloc(h, &G.env, N * 4)) || (rc = dev_alloc(h, &G.tls, N * S * 3)) || (rc = dev_alloc(h, &G.stats, N * ST_N)) ||
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
    set_buf(h, RS_BUF_TLS, G.tls, RS_I32, 3, n, s, 3);
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

    h->lds = lds_bytes_for(C, h->K.n_cells, h->K.n_arr, h->K.n_dep, sc->n_obs, sc->n_signals, sc->n_vtypes, tls_maxl);
    if (h->lds > 160 * 1024) { h->err = "scenario needs more than 160 KiB of LDS per environment"; return fail(RS_ELIMIT); }
    if (block_threads <= 0) {
        block_threads = C >= 512 ? 512 : (C >= 256 ? 256 : (C >= 128 ? 128 : 64));
    }
    if (block_threads % 64 || block_threads > 1024 || block_threads < 64) { h->err = "block_threads must be a multiple of 64 in [64, 1024]"; return fail(RS_EINVAL); }
    h->block = block_threads;
    {
        // the dynamic-LDS ceiling is an attribute of the kernel (per device), not of a launch: only ever raise it,
        // or a handle created earlier for a larger scenario could no longer launch
        static std::mutex mu;
        static size_t max_lds[64] = {0};
        std::lock_guard<std::mutex> lock(mu);
        size_t &cur = max_lds[device_id & 63];
        if (h->lds > cur) {
            for (int c : kStepCaps)        // every instantiation: the ceiling is per kernel function
                if (hipFuncSetAttribute((const void *)step_kernel_for(c), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds) != hipSuccess) {
                    h->err = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed"; return fail(RS_EHIP);
                }
            cur = h->lds;
        }
    }
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { h->err = "hipStreamCreate failed"; return fail(RS_EHIP); }
    *out = h;
    int r2 = rs_reset(h, nullptr);
    if (r2) { g_create_err = h->err; *out = nullptr; rs_destroy(h); return r2; }
    return RS_OK;
}

extern "C" void rs_destroy(rs_handle h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
#ifdef RS_BARWAIT
    {
        unsigned long long b[64];
        int ln[56];
        (void)hipDeviceSynchronize();
        if (hipMemcpyFromSymbol(b, HIP_SYMBOL(g_bar), sizeof(b)) == hipSuccess && hipMemcpyFromSymbol(ln, HIP_SYMBOL(g_bar_line), sizeof(ln)) == hipSuccess) {
            fprintf(stderr, "RS_BARWAIT barrier wait %llu  wave lifetime %llu  (%.1f %%)  barriers per wave %llu\n", b[0], b[1], 100.0 * (double)b[0] / (double)(b[1] ? b[1] : 1), b[2]);
            for (int i = 0; i < 32; ++i) if (b[8 + i]) fprintf(stderr, "RS_BARWAIT   barrier at resco_kernels.h:%d  %.1f %% of the wave lifetime\n", ln[i], 100.0 * (double)b[8 + i] / (double)(b[1] ? b[1] : 1));
        }
    }
#endif
#ifdef RS_COUNT
    {
        unsigned long long c[32];
        (void)hipDeviceSynchronize();
        if (hipMemcpyFromSymbol(c, HIP_SYMBOL(g_count), sizeof(c)) == hipSuccess) {
            static const char *nm[] = {"leader_of calls", "leader_of own-cell nodes", "leader_of further cells", "leader_of further nodes", "rearmost calls",
                                       "rearmost cells", "rearmost nodes", "neighbours calls", "neighbours nodes", "list_push calls", "list_push CAS rounds",
                                       "hop iterations", "foe records", "mover-flag cells", "move lane hand-overs", "choose_link records", "lane-change candidates",
                                       "vehicle-ticks (plan)", "vehicles entering the hop loop"};
            for (int i = 0; i < 19; ++i) fprintf(stderr, "RS_COUNT %-32s %llu\n", nm[i], c[i]);
        }
    }
#endif
    if (h->stream) { (void)wait_idle(h); (void)hipStreamDestroy(h->stream); }
    for (auto &e : h->events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (void *p : h->allocs) (void)hipFree(p);
    delete h;
}

extern "C" const char *rs_last_error(rs_handle h) { return h ? h->err.c_str() : g_create_err.c_str(); }

static int launch_step(rs_sim *h, hipStream_t st, int n_ticks, int do_fsm) {
    KParams P = h->P;
    P.n_ticks = n_ticks; P.do_fsm = do_fsm; P.prof = h->prof;
#ifdef RS_DIAG
    {   // skip mask applied only after RS_DIAG_AFTER launches, so that the traffic state is the real one
        static int n_launch = 0;
        const char *e = getenv("RS_DIAG_SKIP"), *a = getenv("RS_DIAG_AFTER");
        P.diag = (e && ++n_launch > (a ? atoi(a) : 0)) ? atoi(e) : 0;
    }
#endif
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
    hipLaunchKernelGGL(step_kernel_for(h->K.capacity), dim3(h->n_envs), dim3(h->block), h->lds, st, h->K, h->G, h->O, P, (const int32_t *)h->actions);
    HIPCHK(h, hipGetLastError());
    if (h->timing) HIPCHK(h, hipEventRecord(e1, st));
    return RS_OK;
}

extern "C" int rs_reset(rs_handle h, void *stream) {
    if (!h) return RS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    h->last = st;
    hipLaunchKernelGGL(rs_reset_kernel, dim3(h->n_envs), dim3(256), 0, st, h->T, h->G, h->P);
    HIPCHK(h, hipGetLastError());
    bool tm = h->timing;
    h->timing = false;
    int rc = launch_step(h, st, 0, 0);
    h->timing = tm;
    return rc;
}

extern "C" int rs_step(rs_handle h, const int32_t *actions, int32_t actions_on_device, void *stream) {
    if (!h) return RS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    h->last = st;
    if (actions) {
        size_t bytes = (size_t)h->n_envs * h->T.n_signals * sizeof(int32_t);
        HIPCHK(h, hipMemcpyAsync(h->actions, actions, bytes, actions_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
        // a pageable host source may be read after the call returns: make the caller's buffer reusable
        if (!actions_on_device) HIPCHK(h, hipStreamSynchronize(st));
    }
    return launch_step(h, st, h->T.step_length, 1);
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
    int total = h->n_envs * h->T.n_signals;
    hipLaunchKernelGGL(rs_act_random_kernel, dim3((total + 255) / 256), dim3(256), 0, st, h->T, h->P, step_key, h->actions);
    HIPCHK(h, hipGetLastError());
    return RS_OK;
}

extern "C" int rs_act_maxwave(rs_handle h, const int32_t *phase_pairs, int32_t n_pairs, const int32_t *valid,
                              const int32_t *order, int32_t use_pressure, void *stream) {
    if (!h || n_pairs <= 0) return RS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    h->last = st;
    if (!h->pairs) {
        if (!phase_pairs || !valid || !order) { h->err = "rs_act_maxwave: tables required on first use"; return RS_EINVAL; }
        int rc;
        if ((rc = dev_alloc(h, &h->pairs, (size_t)n_pairs * 2, false)) || (rc = dev_alloc(h, &h->valid, (size_t)h->T.n_signals * n_pairs, false)) ||
            (rc = dev_alloc(h, &h->order, (size_t)h->T.n_signals * n_pairs, false))) return rc;
        HIPCHK(h, hipMemcpy(h->order, order, (size_t)h->T.n_signals * n_pairs * 4, hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(h->pairs, phase_pairs, (size_t)n_pairs * 2 * 4, hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(h->valid, valid, (size_t)h->T.n_signals * n_pairs * 4, hipMemcpyHostToDevice));
        h->n_pairs = n_pairs;
    }
    int total = h->n_envs * h->T.n_signals;
    hipLaunchKernelGGL(rs_act_maxwave_kernel, dim3((total + 255) / 256), dim3(256), 0, st, h->T, h->P, (const int32_t *)h->pairs,
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

struct Snapshot { std::vector<void *> ptrs; };
// state AND the observation buffers: re-running observe would advance Signal.waiting_times
static const int kSnapBufs[] = {RS_BUF_LANE_AGG, RS_BUF_DRQ_NORM, RS_BUF_PHASE, RS_BUF_MPLIGHT, RS_BUF_WAVE, RS_BUF_WAIT,
                                RS_BUF_WAIT_NORM, RS_BUF_PRESSURE, RS_BUF_QUEUE_SUM, RS_BUF_QUEUE_MAX, RS_BUF_DRQ_NORM_F16,
                                RS_BUF_ENV, RS_BUF_TLS, RS_BUF_VEH_POS, RS_BUF_VEH_SPEED, RS_BUF_VEH_ACCEL, RS_BUF_VEH_TLOSS,
                                RS_BUF_VEH_LANE, RS_BUF_VEH_TRIP, RS_BUF_VEH_CURSOR, RS_BUF_VEH_SWAIT, RS_BUF_VEH_RWAIT,
                                RS_BUF_VEH_DEPART, RS_BUF_VEH_OWNER, RS_BUF_VEH_SF, RS_BUF_VEH_WTOT, RS_BUF_TRIP_LOG, RS_BUF_STATS};
extern "C" int rs_snapshot(rs_handle h, void **snap) {
    if (!h || !snap) return RS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, wait_idle(h));
    Snapshot *S = new Snapshot();
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
    if (max_lanes_per_signal) *max_lanes_per_signal = h->T.lmax;
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
        (rc = pol_upload<int32_t>(p, &p->W.n_actions, n_actions, S))) {
        g_create_err = "rs_idqn_create: device allocation / upload failed";
        rs_idqn_destroy(p);
        return rc;
    }
    *out = p;
    return RS_OK;
}

extern "C" int rs_idqn_act(rs_policy_handle p, const void *obs, int32_t n_envs, int32_t mode, float epsilon, uint32_t seed, uint32_t step_key,
                           const void *dyn, int32_t *actions, float *q, void *stream) {
    if (!p || !obs || !actions || n_envs <= 0 || mode < 0 || mode > 1) return RS_EINVAL;
    if (hipSetDevice(p->device) != hipSuccess) return RS_EHIP;
    typedef void (*pol_fn)(PolicyTab, const __half *, int, int, float, uint32_t, uint32_t, const uint32_t *, int32_t *, float *);
    static const pol_fn kernels[9] = {nullptr, rs_idqn_forward_kernel<1>, rs_idqn_forward_kernel<2>, rs_idqn_forward_kernel<3>,
                                      rs_idqn_forward_kernel<4>, rs_idqn_forward_kernel<5>, rs_idqn_forward_kernel<6>,
                                      rs_idqn_forward_kernel<7>, rs_idqn_forward_kernel<8>};
    hipLaunchKernelGGL(kernels[p->W.hp], dim3((n_envs + POL_TM - 1) / POL_TM, p->W.S), dim3(128), 0, (hipStream_t)stream,
                       p->W, (const __half *)obs, (int)n_envs, (int)mode, epsilon, seed, step_key, (const uint32_t *)dyn, actions, q);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_EHIP;
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

extern "C" void rs_idqn_destroy(rs_policy_handle p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();
    for (void *d : p->allocs) (void)hipFree(d);
    delete p;
}
