// resco_tables.h -- packed scenario records of the step kernel and the host-side builder that derives them from the
// flat rs_scenario arrays (include/resco_sim.h).  Plain C++ (no HIP): resco_sim.hip uploads the vectors built here, and
// the host emulation of the kernel used by the CPU tests (tests/hostemu) reads them in place.
#pragma once
#include <stdint.h>
#include <string.h>
#include <string>
#include <vector>

#include "resco_model.h"
#include "resco_sim.h"

enum { VT_LENGTH, VT_MINGAP, VT_ACCEL, VT_DECEL, VT_TAU, VT_SIGMA, VT_MAXSPEED, VT_SF_MEAN, VT_SF_DEV, VT_EMERGENCY, VT_COLS };
enum { TLS_R = 0, TLS_Y = 1, TLS_g = 2, TLS_G = 3 };
enum { ST_INSERTED, ST_ARRIVED, ST_DURATION, ST_DEPDELAY, ST_WAITING, ST_TLOSS, ST_ACTIVE, ST_PENDING, ST_ACTIVE_TICKS, ST_TICKS, ST_CAP_BLOCKED, ST_INVARIANT, ST_N };

#define LANE_NONE 0xFFFFu
#define OWNER_NONE 0xFFu
#define NIL 0x7FF               /* empty grid cell / end of a cell chain (11-bit slot ids) */
// a grid cell (16 bits): bits 0..10 the head slot of its chain (NIL: empty), bits 11..13 the number of vehicles in it
// (saturating at 7: the cell length is chosen so that a cell cannot hold more fronts, see PackedTables::build), bit 14 = it holds a
// moving vehicle, bit 15 = the TAG: the parity of the tick the contents are valid for.  There is ONE grid: the move of tick t pushes
// the vehicles into it with the tag of tick t + 1, and whoever reads or pushes takes a cell that carries the other tag for empty
// (resco_step.h: Grid).  Round 6: the second grid of rounds 1-5 (9 KB of working memory for ingolstadt21) is what stood between
// three and four workgroups per CU.
#define CELL_CNT_SHIFT 11
#define CELL_CNT_MAX 7u
#define CELL_MOVER 0x4000u
#define CELL_TAG 0x8000u
#define ARR_NONE 65535
#define COOP_NONE 0xFFFFFFFFu
#define TRIP_NONE 0xFFFFu

// The lanes are covered by a grid of CELL_LEN-metre cells (floor(len / CELL_LEN) + 1 per lane, lanes in index order):
// a cell holds the slot of a vehicle whose front is inside it (more than one: a short chain).  Every neighbour search of
// the model is a bounded scan over a few consecutive cells.
// The cell length is chosen per scenario when a handle is created (pick_cell_len, resco_step.h): the shortest of CELL_CHOICES with which the
// working memory of an environment lets FOUR workgroups share a CU -- or, failing that, three -- (resco_tables.h); shorter cells mean
// shorter chains (30 m instead of 32 m: +1.8 % on ingolstadt21 x 4096, profiles/r05_ab_cells.txt), one workgroup less per CU costs
// 15-25 %.
#ifdef CELL_LEN                 // (study builds: one fixed length)
#define CELL_CHOICES {CELL_LEN}
#else
#define CELL_CHOICES {30.0f, 32.0f, 36.0f, 40.0f, 48.0f}
#endif
// bytes of LDS per workgroup up to which three / four of them fit one CU of the MI355X: 160 KiB in granules of 1280 bytes (measured in
// round 5: 53 664 three, 53 824 two; round 6: 40 960 = 32 granules four)
#define RS_LDS_3WG_LIMIT 53760
#define RS_LDS_4WG_LIMIT 40960

// ---- 16-byte records: one global_load_dwordx4 fetches everything about a lane / foe / route
struct __attribute__((aligned(16))) LaneRec {
    float len, vmax;
    uint16_t link_start;
    uint8_t link_cnt;
    uint8_t flags;          // bit0 junction-internal; bits 2..7 number of lanes of the edge
    uint16_t cell0;         // first grid cell of this lane
    uint16_t edge_lane0;
};
struct __attribute__((aligned(16))) LinkRec {
    uint16_t to_lane, to_edge, foe_start, via2;     // via2 0xFFFF: none
    int16_t arr_idx;                                // approach register of this link (only foe targets have one)
    uint8_t tls, tls_pos;                           // tls 0xFF: uncontrolled
    uint8_t foe_cnt, flags;                         // flags bit0 minor, bit1 cont
    uint8_t dest_k, pad;                            // lane index of the destination lane inside to_edge
    LaneRec dest;                                   // copy of lanes[to_lane]: one dependent gather less per hop
};
struct __attribute__((aligned(16))) FoeRec {
    int16_t arr_idx;
    uint8_t tls, tls_pos;
    uint16_t via1_cell0, via2_cell0;    // first cell of the foe's junction lanes (0xFFFF: none)
    uint8_t via1_nc, via2_nc, pad[6];   // number of cells of those lanes
};
struct __attribute__((aligned(8))) RStep {
    uint16_t edge, next_edge;           // next_edge 0xFFFF: last edge of the route
    float tlsdist;
};
struct __attribute__((aligned(16))) RouteRec {
    uint32_t start;
    uint16_t depart_lane;
    int16_t dep_idx;            // index of the departure lane among the departure lanes (insertion backlog)
    uint16_t depart_cell0;
    uint16_t pad;
    float depart_len;
};
#define LF_INTERNAL 1u
#define KF_MINOR 1u
#define KF_CONT 2u
#define NLINK_NONE 0x7FFF       // L.nlink value: link index (0x7FFF none), bit 15 = the link owns an approach register
#define NLINK_ARR 0x8000

struct DepInfo { float len; uint16_t cell0, lane; };
// tables used rarely (per signal, per departure lane, at load / observe): reached through one pointer
struct KCold {
    const int32_t *trip_depart;
    const uint16_t *trip_next;          // next trip with the same departure lane (TRIP_NONE: last)
    const uint16_t *dep_lane;           // [n_dep] lane of each departure lane, ascending
    const struct DepInfo *dep_info;     // [n_dep] what the insertion check needs to know about that lane: one 8-byte load
    const uint16_t *dep_first;          // [n_dep] its first trip
    const float *vtype_params;
    const uint8_t *tls8, *fix8;
    const int32_t *tls_nphase, *tls_ngreen, *tls_nlinks, *tls_state_off, *tls_dur_off, *tls_yel_off, *tls_dur, *tls_yellow, *tls_init_phase;
    const int32_t *fix_nphase, *fix_state_off, *fix_dur_off, *fix_dur, *fix_init_phase, *fix_init_left;
    const int16_t *lane_obs;
    const int32_t *obs_sig, *sig_obs_start, *mv_in_start, *mv_in_idx, *mv_out_start, *mv_out_idx, *pr_out_start, *pr_out_idx;
    const int32_t *trips_cum;
};
// tables of the per-vehicle, per-tick path: by value (SGPRs)
// RS_G(p): "p points to global memory" -- the HIP build launders the pointer through the global address space so that
// table accesses become global_load (a pointer LOADED from the constant argument block is otherwise generic and every
// access a flat one, which also ties up the LDS wait counter); the host emulation defines it as the identity.
#ifndef RS_G
#define RS_G(p) (p)
#endif
#ifndef RS_MEM
#define RS_MEM inline
#endif
struct KTab {
    const LaneRec *lanes_;
    const LinkRec *links_;
    const FoeRec *foes_;
    const RStep *rsteps_;
    const float *route_cont_;       // [n_route_steps][kmax]
    const uint16_t *next_link_;     // [n_route_steps][kmax][2]: choose_link() of a normal lane for even / odd trips, NLINK_NONE: none
    const uint16_t *notbest_;       // [n_route_steps] bit k: lane k of the step's edge is not one of its best lanes (= the signs of route_cont's row)
    const RouteRec *routes_;
    const uint16_t *trip_route_;
    const uint8_t *trip_vtype_;
    KCold cold;                     // by value: every table pointer is then loaded from the constant argument block
    float maxlen;
    float occ_unit;                 // length + minGap of the most common vehicle type (oracle: occ_unit)
    int32_t n_trips, tls_maxl, kmax;
    int32_t n_lanes, n_cells, n_signals, n_obs, n_vtypes, horizon, capacity, step_length, yellow_length, lmax, n_arr, n_dep;
    RS_MEM const LaneRec *lanes() const { return RS_G(lanes_); }
    RS_MEM const LinkRec *links() const { return RS_G(links_); }
    RS_MEM const FoeRec *foes() const { return RS_G(foes_); }
    RS_MEM const RStep *rsteps() const { return RS_G(rsteps_); }
    RS_MEM const float *route_cont() const { return RS_G(route_cont_); }
    RS_MEM const uint16_t *next_link() const { return RS_G(next_link_); }
    RS_MEM const uint16_t *notbest() const { return RS_G(notbest_); }
    RS_MEM const RouteRec *routes() const { return RS_G(routes_); }
    RS_MEM const uint16_t *trip_route() const { return RS_G(trip_route_); }
    RS_MEM const uint8_t *trip_vtype() const { return RS_G(trip_vtype_); }
};

struct KParams {
    uint32_t seed;
    int32_t env_base;
    float max_distance, sigma;
    int32_t speed_dev, fixed_program;
    int32_t tls_expiry;     // !rs_params.tls_hold: 1 = a phase set through setPhase expires after its programme duration (SUMO's setPhase, the default)
    int32_t n_ticks;        // ticks to simulate in this launch (0: observe only)
    int32_t do_fsm;         // apply prep_phase / set_phase around the ticks
    int32_t do_observe;     // 1: Signal.observe + states / rewards after the ticks; 0: only the state goes back (step_sim)
    uint32_t out_mask;      // which of the per-lane / per-movement output buffers an observe writes, see OUT_*
    int32_t n_envs;
    unsigned long long *prof;   // optional [16] per-phase cycle accumulators (rs_phase_profile), NULL = off
};

// output groups (KParams.out_mask); the per-signal scalars (phase, rewards, queue metrics, arrivals / departures) are always written
#define OUT_LANE_AGG 1u
#define OUT_DRQ_NORM 2u
#define OUT_DRQ_F16 4u
#define OUT_LANE_ARR 8u
#define OUT_MPLIGHT 16u
#define OUT_WAVE 32u
#define OUT_MPLIGHT_FULL 64u
#define OUT_VEH_ACCEL 128u       // RS_BUF_VEH_ACCEL (the last tick's acceleration per vehicle: only the Signal views' vehicle dicts read it)
#define OUT_ALL 255u
#define TLS_W 4         // ints per signal in State.tls: phase, time left, next_phase, |Signal.departures| collected since the last observe

// ---------------------------------------------------------------------------------------------- host-side builder
struct PackedTables {
    std::vector<LaneRec> lanes;
    std::vector<LinkRec> links;
    std::vector<FoeRec> foes;
    std::vector<RStep> rsteps;
    std::vector<RouteRec> routes;
    std::vector<uint16_t> next_link, trip_route, trip_next, dep_lane, dep_first, notbest;
    std::vector<uint8_t> trip_vtype, tls8, fix8;
    std::vector<DepInfo> dep_info;
    std::vector<int32_t> tls_off_p, fix_off_p;      // byte offset of signal s's first row in tls8 / fix8 (rows padded to tls_maxl bytes)
    std::vector<int16_t> lane_obs16;
    std::vector<int32_t> obs_sig;
    std::vector<float> route_cont;          // sc->route_cont with the sign bit set where the lane is not one of the best of its route step (resco_step.h: cont_notbest)
    int n_cells = 0, n_arr = 1, n_dep = 1, kmax = 1, lmax = 1, tls_maxl = 1;
    float maxlen = 0.0f, occ_unit = 0.0f;
    float cell_len = 30.0f, cell_inv = 1.0f / 30.0f;       // grid cell length of this build of the tables (pick_cell_len)
    std::string err;

    // the link a vehicle on normal lane `ln` takes towards route step q + 1 (oracle/resco_oracle.c choose_link restated
    // as a table): of the connections to the next edge those whose destination lane is the best one or can be followed
    // RM_GOOD_CONT metres are acceptable; even trips take the first of them, odd trips the second (if there is one)
    static void choose(const rs_scenario *sc, int q, int ln, int out[2]) {
        out[0] = out[1] = -1;
        const int ne = sc->route_edge[q + 1];
        const float *cn = sc->route_cont + (size_t)(q + 1) * sc->kmax;
        float bc = -1.0f;
        for (int l = sc->lane_link_start[ln]; l < sc->lane_link_start[ln] + sc->lane_link_cnt[ln]; ++l) {
            if (sc->link_to_edge[l] != ne) continue;
            const float c = cn[sc->link_dest_lane[l] - sc->edge_lane0[ne]];
            if (c > bc) bc = c;
        }
        if (bc < RM_MIN_LC_LEN) return;
        int n = 0;
        for (int l = sc->lane_link_start[ln]; l < sc->lane_link_start[ln] + sc->lane_link_cnt[ln] && n < 2; ++l) {
            if (sc->link_to_edge[l] != ne) continue;
            const float c = cn[sc->link_dest_lane[l] - sc->edge_lane0[ne]];
            if (c >= bc - RM_CONT_EPS || c >= RM_GOOD_CONT) out[n++] = l;
        }
        if (n == 1) out[1] = out[0];
    }

    // grid cells of the whole network for a given cell length (what build() lays out)
    static int count_cells(const rs_scenario *sc, float len) {
        const float inv = 1.0f / len;
        int n = 0;
        for (int l = 0; l < sc->n_lanes; ++l) n += (int)(sc->lane_len[l] * inv) + 1;
        return n;
    }

    // can a grid of `len`-metre cells count the vehicles of this scenario?  (the 3-bit counter of a cell saturates at 7, the oracle's
    // per-lane count is exact: a vehicle type must not be so short that a cell could hold more fronts than that -- in a standing
    // queue, length + minGap apart, with one to spare, and bumper to bumper, what an urgent lane change accepts, at all)
    static bool cell_len_ok(const rs_scenario *sc, float len_) {
        for (int v = 0; v < sc->n_vtypes; ++v) {
            const float len = sc->vtype_params[v * VT_COLS + VT_LENGTH], unit = len + sc->vtype_params[v * VT_COLS + VT_MINGAP];
            if (!(len > 0.0f) || (int)(len_ / unit) + 1 >= (int)CELL_CNT_MAX || (int)(len_ / len) + 1 > (int)CELL_CNT_MAX) return false;
        }
        return true;
    }

    bool build(const rs_scenario *sc, float cell_len_ = 30.0f) {
        cell_len = cell_len_; cell_inv = 1.0f / cell_len_;
        if (sc->n_lanes >= 0xFFFE || sc->n_trips >= 0xFFFF || sc->n_routes > 0xFFFF || sc->n_vtypes > 255 || sc->n_signals > 254) {
            err = "scenario exceeds id widths (lanes/trips/routes u16, vtypes/signals u8)"; return false;
        }
        if (sc->n_route_steps >= 0xFFFF || sc->n_foes >= 0xFFFF || sc->n_links >= 0x7FFF || sc->n_edges >= 0xFFFF || sc->n_obs >= 0x7FFF) {
            err = "scenario exceeds packed-table id widths (route steps / foes / links / edges u16)"; return false;
        }
        obs_sig.assign((size_t)(sc->n_obs > 0 ? sc->n_obs : 1), 0);
        for (int s = 0; s < sc->n_signals; ++s) {
            for (int oi = sc->sig_obs_start[s]; oi < sc->sig_obs_start[s + 1]; ++oi) obs_sig[oi] = s;
            const int n = sc->sig_obs_start[s + 1] - sc->sig_obs_start[s];
            if (n > lmax) lmax = n;
            if (sc->tls_nlinks[s] > tls_maxl) tls_maxl = sc->tls_nlinks[s];
        }
        for (int v = 0; v < sc->n_vtypes; ++v)
            if (sc->vtype_params[v * VT_COLS + VT_LENGTH] > maxlen) maxlen = sc->vtype_params[v * VT_COLS + VT_LENGTH];
        {   // what one queued vehicle occupies: the most common vehicle type (ties: the lower index)
            std::vector<int> cnt((size_t)(sc->n_vtypes > 0 ? sc->n_vtypes : 1), 0);
            for (int k = 0; k < sc->n_trips; ++k) cnt[sc->trip_vtype[k]] += 1;
            int best = 0;
            for (int v = 1; v < sc->n_vtypes; ++v) if (cnt[v] > cnt[best]) best = v;
            occ_unit = (sc->vtype_params[best * VT_COLS + VT_LENGTH] + sc->vtype_params[best * VT_COLS + VT_MINGAP]) * RM_OCC_FACTOR;
        }
        if (sc->capacity >= NIL) { err = "capacity exceeds the 11-bit slot ids of the grid cells"; return false; }
        if (!cell_len_ok(sc, cell_len)) {
            err = "a vehicle type is too short for the grid cells: floor(cell length / (length + minGap)) + 1 must stay below 7 (and floor(cell length / length) + 1 at or below 7)"; return false;
        }
        std::vector<int16_t> link_arr((size_t)sc->n_links, -1);
        int n_foe_targets = 0;
        for (int l = 0; l < sc->n_links; ++l)
            for (int i = sc->link_foe_start[l]; i < sc->link_foe_start[l] + sc->link_foe_cnt[l]; ++i) {
                const int f = sc->foe_link[i];
                if (link_arr[f] < 0) link_arr[f] = (int16_t)n_foe_targets++;
            }
        n_arr = n_foe_targets > 0 ? n_foe_targets : 1;
        lanes.resize((size_t)sc->n_lanes);
        std::vector<int> lane_nc((size_t)sc->n_lanes);
        for (int l = 0; l < sc->n_lanes; ++l) {
            LaneRec &R = lanes[l];
            R.len = sc->lane_len[l]; R.vmax = sc->lane_vmax[l];
            R.link_start = (uint16_t)sc->lane_link_start[l];
            if (sc->lane_link_cnt[l] > 255) { err = "more than 255 links on one lane"; return false; }
            R.link_cnt = (uint8_t)sc->lane_link_cnt[l];
            const int e = sc->lane_edge[l];
            const int nl = e >= 0 ? sc->edge_nlanes[e] : 0;
            R.flags = (uint8_t)((sc->lane_internal[l] ? LF_INTERNAL : 0u) | ((unsigned)nl << 2));
            R.edge_lane0 = (uint16_t)(e >= 0 ? sc->edge_lane0[e] : 0);
            // grid cells: lanes of one edge are consecutive and equally long, so their cell blocks are consecutive and
            // equally sized (relied on by the lane change: the neighbour lane's block is one block further)
            lane_nc[l] = (int)(sc->lane_len[l] * cell_inv) + 1;
            if (lane_nc[l] > 255) { err = "lane longer than 255 grid cells"; return false; }
            R.cell0 = (uint16_t)n_cells;
            n_cells += lane_nc[l];
        }
        if (n_cells >= 0xFFF0) { err = "too many grid cells"; return false; }
        for (int e = 0; e < sc->n_edges; ++e)
            for (int j = 1; j < sc->edge_nlanes[e]; ++j)
                if (lane_nc[sc->edge_lane0[e] + j] != lane_nc[sc->edge_lane0[e]]) { err = "lanes of one edge differ in length"; return false; }
        lane_obs16.resize((size_t)sc->n_lanes);
        for (int l = 0; l < sc->n_lanes; ++l) lane_obs16[l] = (int16_t)sc->lane_obs[l];
        links.resize((size_t)sc->n_links);
        for (int l = 0; l < sc->n_links; ++l) {
            LinkRec &R = links[l];
            R.to_lane = (uint16_t)sc->link_to_lane[l]; R.to_edge = (uint16_t)sc->link_to_edge[l];
            R.foe_start = (uint16_t)sc->link_foe_start[l];
            R.via2 = sc->link_via2[l] >= 0 ? (uint16_t)sc->link_via2[l] : (uint16_t)0xFFFF;
            R.arr_idx = link_arr[l];
            R.tls = sc->link_tls[l] >= 0 ? (uint8_t)sc->link_tls[l] : (uint8_t)0xFF;
            R.tls_pos = sc->link_tls[l] >= 0 ? (uint8_t)sc->link_tls_pos[l] : (uint8_t)0;
            if (sc->link_foe_cnt[l] > 255 || (sc->link_tls[l] >= 0 && sc->link_tls_pos[l] > 255)) { err = "foe count / TLS link index exceeds u8"; return false; }
            R.foe_cnt = (uint8_t)sc->link_foe_cnt[l];
            R.flags = (uint8_t)((sc->link_minor[l] ? KF_MINOR : 0u) | (sc->link_cont[l] ? KF_CONT : 0u));
            R.dest_k = (uint8_t)(sc->link_dest_lane[l] - sc->edge_lane0[sc->link_to_edge[l]]);
            R.pad = 0;
            R.dest = lanes[sc->link_to_lane[l]];
        }
        foes.resize((size_t)(sc->n_foes > 0 ? sc->n_foes : 1));
        for (int i = 0; i < sc->n_foes; ++i) {
            const int f = sc->foe_link[i];
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
        kmax = sc->kmax;
        {   // continuation lengths, signed: lane k of route step q is "not a best lane" iff !(c_k >= max_j c_j - RM_CONT_EPS) over the lanes of
            // the step's edge -- the very comparison the oracle's strategic_dir_at makes per vehicle and tick
            const size_t nq = (size_t)(sc->n_route_steps > 0 ? sc->n_route_steps : 1);
            route_cont.assign(nq * kmax, 0.0f);
            notbest.assign(nq, 0);
            for (int q = 0; q < sc->n_route_steps; ++q) {
                const float *c = sc->route_cont + (size_t)q * kmax;
                const int n = sc->edge_nlanes[sc->route_edge[q]];
                float best = 0.0f;
                for (int j = 0; j < n && j < kmax; ++j) if (c[j] > best) best = c[j];
                for (int j = 0; j < kmax; ++j) {
                    float v = c[j];
                    if (!(v >= 0.0f)) { err = "negative continuation length"; return false; }
                    if (j < n && !(v >= best - RM_CONT_EPS)) { uint32_t b; memcpy(&b, &v, 4); b |= 0x80000000u; memcpy(&v, &b, 4); notbest[(size_t)q] |= (uint16_t)(1u << j); }
                    route_cont[(size_t)q * kmax + j] = v;
                }
            }
        }
        rsteps.resize((size_t)(sc->n_route_steps > 0 ? sc->n_route_steps : 1));
        routes.resize((size_t)sc->n_routes);
        next_link.assign((size_t)(sc->n_route_steps > 0 ? sc->n_route_steps : 1) * kmax * 2, (uint16_t)NLINK_NONE);
        std::vector<int16_t> lane_dep((size_t)sc->n_lanes, -1);
        // departure lanes [SUMO-K departLane "first"]: the right-most lane of a route's first edge, numbered in lane order
        for (int r = 0; r < sc->n_routes; ++r) lane_dep[sc->edge_lane0[sc->route_edge[sc->route_start[r]]]] = 0;
        dep_lane.clear();
        for (int l = 0; l < sc->n_lanes; ++l) if (lane_dep[l] == 0) { lane_dep[l] = (int16_t)dep_lane.size(); dep_lane.push_back((uint16_t)l); }
        n_dep = dep_lane.empty() ? 1 : (int)dep_lane.size();
        if (dep_lane.empty()) dep_lane.push_back(0);
        for (int r = 0; r < sc->n_routes; ++r) {
            const int rs = sc->route_start[r], re = sc->route_start[r + 1];
            for (int q = rs; q < re; ++q) {
                RStep &R = rsteps[q];
                R.edge = (uint16_t)sc->route_edge[q];
                const bool last = q + 1 >= re;
                R.next_edge = last ? (uint16_t)0xFFFF : (uint16_t)sc->route_edge[q + 1];
                R.tlsdist = sc->route_tlsdist[q];
                if (last) continue;
                const int e = sc->route_edge[q];
                for (int k = 0; k < sc->edge_nlanes[e]; ++k) {
                    int alt[2];
                    choose(sc, q, sc->edge_lane0[e] + k, alt);
                    for (int j = 0; j < 2; ++j)
                        if (alt[j] >= 0) next_link[((size_t)q * kmax + k) * 2 + j] = (uint16_t)(alt[j] | (link_arr[alt[j]] >= 0 ? NLINK_ARR : 0));
                        else next_link[((size_t)q * kmax + k) * 2 + j] = (uint16_t)NLINK_NONE;
                }
            }
            const int dl = sc->edge_lane0[sc->route_edge[rs]];
            routes[r].start = (uint32_t)rs; routes[r].depart_lane = (uint16_t)dl; routes[r].dep_idx = lane_dep[dl];
            routes[r].depart_cell0 = lanes[dl].cell0; routes[r].pad = 0; routes[r].depart_len = sc->lane_len[dl];
        }
        trip_route.resize((size_t)sc->n_trips); trip_vtype.resize((size_t)sc->n_trips); trip_next.assign((size_t)sc->n_trips, (uint16_t)TRIP_NONE);
        dep_first.assign((size_t)n_dep, (uint16_t)TRIP_NONE);
        for (int k = sc->n_trips - 1; k >= 0; --k) {
            trip_route[k] = (uint16_t)sc->trip_route[k]; trip_vtype[k] = (uint8_t)sc->trip_vtype[k];
            const int d = routes[sc->trip_route[k]].dep_idx;
            trip_next[k] = dep_first[d];
            dep_first[d] = (uint16_t)k;
        }
        // link states per (signal, phase): one row of tls_maxl bytes (a multiple of 4, zero padded) per phase, so that a phase
        // change copies a row into the working memory with a handful of independent 32-bit loads instead of one byte at a time
        tls_maxl = (tls_maxl + 3) & ~3;
        dep_info.clear();
        for (size_t d = 0; d < dep_lane.size(); ++d) dep_info.push_back(DepInfo{lanes[dep_lane[d]].len, lanes[dep_lane[d]].cell0, dep_lane[d]});
        tls_off_p.assign((size_t)sc->n_signals, 0); fix_off_p.assign((size_t)sc->n_signals, 0);
        tls8.clear(); fix8.clear();
        for (int s = 0; s < sc->n_signals; ++s) {
            const int n = sc->tls_nlinks[s];
            tls_off_p[(size_t)s] = (int32_t)tls8.size();
            for (int ph = 0; ph < sc->tls_nphase[s]; ++ph)
                for (int i = 0; i < tls_maxl; ++i) tls8.push_back(i < n ? (uint8_t)sc->tls_states[sc->tls_state_off[s] + ph * n + i] : (uint8_t)0);
            fix_off_p[(size_t)s] = (int32_t)fix8.size();
            for (int ph = 0; ph < sc->fix_nphase[s]; ++ph)
                for (int i = 0; i < tls_maxl; ++i) fix8.push_back(i < n ? (uint8_t)sc->fix_states[sc->fix_state_off[s] + ph * n + i] : (uint8_t)0);
        }
        tls8.resize(tls8.size() + 64, 0); fix8.resize(fix8.size() + 64, 0);       // (a row copy reads whole dwords up to 32 bytes)
        return true;
    }
};
