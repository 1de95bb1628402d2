/* The oracle's OWN copy of two maps' tables (test infrastructure, like copo_oracle.c).
 *
 * tests/oracle_lib.py feeds the oracle simulator with the product's tables (copo_amd/maps.py through fill_cfg_struct), so
 * "HIP == oracle bit for bit" says nothing about the geometry: both sides would follow a maps.py error together.  This file
 * derives the route / spawn tables of the Intersection and the Roundabout a second time, from MetaDrive 0.2.5's block
 * constants and by a different construction, so that a test can hold the two against each other (tests/test_oracle_golden.py:
 * every field within 2e-4; tests/test_gpu_sim_parity.py: the HIP simulator on maps.py's tables against the oracle on THESE).
 *
 * Constants (MetaDrive 0.2.5; the release's source is not in the reference tree -- README.md:42 names the tag -- so they are
 * restated from the published block definitions, as DESIGN.md section 3.3 says):
 *   lane width 3.5 m, 2 lanes per direction (MAIntersectionConfig / MARoundaboutConfig: map_config lane_num 2, exit_length 60)
 *   FirstPGBlock: the first 10 m are the entrance, the spawn road is the remaining exit_length - 10 = 50 m
 *   InterSection: radius 10 (the rightmost lane's right turn); stop lines at radius + (2 n - 1) w / 2 = 15.25 m from the
 *                 centre; lane-0 (leftmost-lane) radii: right 10 + (n - 1) w = 13.5, left 10 + n w = 17, U-turn w / 2 = 1.75
 *   Roundabout:   radius_exit 10, radius_inner 30, angle 70 deg; radius_big = (2 n - 1) w + radius_inner = 40.5 on the
 *                 rightmost lane; the arc that joins two arms has radius beneath / cos(angle) - radius_exit on the rightmost
 *                 lane with beneath = (2 n - 1) w / 2 + radius_exit (`_create_circular_part`)
 *   SpawnManager: floor(50 / 8) = 6 slots per lane, pitch 50 / 6, the first at 4 m; the first slot of a lane is a respawn place
 *   Navigation:   check point = end of the road, lateral middle ((lanes / 2 - 0.5) w to the right of lane 0); features
 *                 radius / (60 + lanes w), (angle_deg / 135 + 1) / 2, both clipped at 1
 *
 * Construction (NOT maps.py's): the Intersection in closed form about the junction centre -- every arm is arm 0 turned by exact
 * quarter turns (integer cosines), arcs given by their centres, which are the corners of the junction square; the Roundabout as
 * a chain of arcs advanced by ROTATING the start point about the arc's centre (maps.advance uses differences of sines).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../include/copo_hip.h"

#define W 3.5
#define NL 2
#define ROWS (COPO_MAX_SEGS + 1)
#define PI 3.14159265358979323846

typedef struct { double x, y, th; } pose_t;

static double wrap(double a) { a = fmod(a + PI, 2.0 * PI); if (a < 0.0) a += 2.0 * PI; return a - PI; }

/* end pose of `len` metres of curvature `kappa` (left +): the start point turned about the arc's centre */
static pose_t arc_end(pose_t p, double len, double kappa) {
    pose_t q;
    if (kappa == 0.0) { q.x = p.x + cos(p.th) * len; q.y = p.y + sin(p.th) * len; q.th = p.th; return q; }
    double R = 1.0 / fabs(kappa), sg = kappa > 0.0 ? 1.0 : -1.0;
    double cx = p.x - sg * R * sin(p.th), cy = p.y + sg * R * cos(p.th);
    double phi = sg * len / R, c = cos(phi), s = sin(phi), rx = p.x - cx, ry = p.y - cy;
    q.x = cx + c * rx - s * ry; q.y = cy + s * rx + c * ry; q.th = p.th + phi;
    return q;
}
static pose_t shift_left(pose_t p, double d) { pose_t q = {p.x - sin(p.th) * d, p.y + cos(p.th) * d, p.th}; return q; }

typedef struct { pose_t p; double len, kappa; int lsolid, rsolid; } road_t;

/* one road record (include/copo_hip.h: COPO_SEG_*), lane count with the edge-line flags in its fraction */
static void emit(float* rec, const road_t* r, double s0) {
    double v[COPO_SEG_STRIDE];
    memset(v, 0, sizeof(v));
    pose_t e = arc_end(r->p, r->len, r->kappa);
    pose_t ck = shift_left(e, -((double)NL / 2.0 - 0.5) * W);
    v[0] = r->p.x; v[1] = r->p.y; v[2] = cos(r->p.th); v[3] = sin(r->p.th); v[4] = r->len; v[5] = r->kappa; v[6] = s0;
    v[7] = wrap(r->p.th); v[8] = ck.x; v[9] = ck.y; v[10] = (double)NL + 0.25 * r->lsolid + 0.5 * r->rsolid;
    if (r->kappa == 0.0) { v[11] = 0.0; v[12] = 0.0; v[13] = 0.5; v[14] = 1.0; v[15] = 0.0; }
    else {
        double R = 1.0 / fabs(r->kappa), sg = r->kappa > 0.0 ? 1.0 : -1.0, ang = r->len / R;
        double cx = r->p.x - sg * R * sin(r->p.th), cy = r->p.y + sg * R * cos(r->p.th);
        pose_t mid = arc_end(r->p, 0.5 * r->len, r->kappa);
        double fr = R / (60.0 + NL * W), fa = (ang * 180.0 / PI / 135.0 + 1.0) / 2.0;
        v[11] = fr < 1.0 ? fr : 1.0; v[12] = R; v[13] = fa < 1.0 ? fa : 1.0;
        v[14] = (mid.x - cx) / R; v[15] = (mid.y - cy) / R;
    }
    for (int k = 0; k < COPO_SEG_STRIDE; ++k) rec[k] = (float)v[k];
}

static void emit_route(float* segs, float* meta, int route, const road_t* roads, int n) {
    float* base = segs + (size_t)route * ROWS * COPO_SEG_STRIDE;
    double s = 0.0;
    for (int k = 0; k < n; ++k) { emit(base + (size_t)k * COPO_SEG_STRIDE, &roads[k], s); s += roads[k].len; }
    road_t t = roads[n - 1];                       /* terminal records: the end pose, zero length */
    t.p = arc_end(roads[n - 1].p, roads[n - 1].len, roads[n - 1].kappa); t.len = 0.0; t.kappa = 0.0;
    for (int k = n; k < ROWS; ++k) emit(base + (size_t)k * COPO_SEG_STRIDE, &t, s);
    meta[route * 4 + 0] = (float)s; meta[route * 4 + 1] = (float)n; meta[route * 4 + 2] = -1.0f; meta[route * 4 + 3] = 0.0f;
}

static int emit_spawns(int32_t* tab, float* sps, int arms, int per_arm) {
    int P = 0;
    for (int a = 0; a < arms; ++a)
        for (int lane = 0; lane < NL; ++lane)
            for (int j = 0; j < 6; ++j) {
                tab[P * 4 + 0] = a * per_arm; tab[P * 4 + 1] = per_arm; tab[P * 4 + 2] = lane; tab[P * 4 + 3] = j == 0;
                sps[P] = (float)(4.0 + j * (50.0 / 6.0));
                ++P;
            }
    return P;
}

/* a pose given about the junction centre, turned by q exact quarter turns */
static pose_t quarter(double cx, double cy, double x, double y, double th, int q) {
    static const double C[4] = {1.0, 0.0, -1.0, 0.0}, S[4] = {0.0, 1.0, 0.0, -1.0};
    pose_t p = {cx + C[q & 3] * x - S[q & 3] * y, cy + S[q & 3] * x + C[q & 3] * y, th + (q & 3) * (PI / 2.0)};
    return p;
}

static void intersection(float* segs, float* meta, int32_t* tab, float* sps, int32_t* R, int32_t* P) {
    const double half = 10.0 + (2 * NL - 1) * W / 2.0, cx = 60.0 + half, cy = W / 2.0;
    const double r_right = 10.0 + (NL - 1) * W, r_left = 10.0 + NL * W;
    int route = 0;
    for (int a = 0; a < 4; ++a) {
        const double La = a == 0 ? 50.0 : 60.0;
        for (int d = 0; d < 4; ++d) {
            const double Ld = d == 0 ? 50.0 : 60.0;
            road_t rd[3];
            rd[0].p = quarter(cx, cy, -half - La, -W / 2.0, 0.0, a); rd[0].len = La; rd[0].kappa = 0.0; rd[0].lsolid = rd[0].rsolid = 1;
            rd[1].p = quarter(cx, cy, -half, -W / 2.0, 0.0, a); rd[1].lsolid = 0; rd[1].rsolid = 0;
            switch ((d - a + 4) & 3) {
                case 0: rd[1].len = (W / 2.0) * PI; rd[1].kappa = 2.0 / W; break;                     /* U-turn at the stop line */
                case 1: rd[1].len = r_right * PI / 2.0; rd[1].kappa = -1.0 / r_right; rd[1].rsolid = 1; break;
                case 2: rd[1].len = 2.0 * half; rd[1].kappa = 0.0; break;
                default: rd[1].len = r_left * PI / 2.0; rd[1].kappa = 1.0 / r_left; break;
            }
            rd[2].p = quarter(cx, cy, -half, W / 2.0, PI, d); rd[2].len = Ld; rd[2].kappa = 0.0; rd[2].lsolid = rd[2].rsolid = 1;
            emit_route(segs, meta, route++, rd, 3);
        }
    }
    *R = route;
    *P = emit_spawns(tab, sps, 4, 4);
}

static void roundabout(float* segs, float* meta, int32_t* tab, float* sps, int32_t* R, int32_t* P) {
    const double ang = 70.0 * PI / 180.0;
    const double r_e0 = 10.0 + (NL - 1) * W;                                   /* lane 0 of a right bend: outside the rightmost lane's radius */
    const double r_b0 = ((2 * NL - 1) * W + 30.0) - (NL - 1) * W;              /* lane 0 of a left arc: inside the rightmost lane's */
    const double r_j0 = (((2 * NL - 1) * W / 2.0 + 10.0) / cos(ang) - 10.0) - (NL - 1) * W;
    road_t in[4], bend[4], ring[4], out[4], ex[4], join[4];
    pose_t entry = {10.0, 0.0, 0.0};
    for (int a = 0; a < 4; ++a) {
        const int nxt = (a + 1) & 3;
        const double La = a == 0 ? 50.0 : 60.0, Ln = nxt == 0 ? 50.0 : 60.0;
        in[a].p = entry; in[a].len = La; in[a].kappa = 0.0; in[a].lsolid = in[a].rsolid = 1;
        bend[a].p = arc_end(in[a].p, La, 0.0); bend[a].len = r_e0 * ang; bend[a].kappa = -1.0 / r_e0; bend[a].lsolid = 0; bend[a].rsolid = 1;
        ring[a].p = arc_end(bend[a].p, bend[a].len, bend[a].kappa); ring[a].len = r_b0 * (2.0 * ang - PI / 2.0); ring[a].kappa = 1.0 / r_b0;
        ring[a].lsolid = 1; ring[a].rsolid = 0;
        out[a].p = arc_end(ring[a].p, ring[a].len, ring[a].kappa); out[a].len = r_e0 * ang; out[a].kappa = -1.0 / r_e0; out[a].lsolid = 0; out[a].rsolid = 1;
        ex[a].p = arc_end(out[a].p, out[a].len, out[a].kappa); ex[a].len = Ln; ex[a].kappa = 0.0; ex[a].lsolid = ex[a].rsolid = 1;
        join[a].p = out[a].p; join[a].len = r_j0 * (PI - 2.0 * ang); join[a].kappa = 1.0 / r_j0; join[a].lsolid = 1; join[a].rsolid = 0;
        /* the next arm's entry road is the opposite carriageway of this exit: lane 0 one lane width to the left, reversed */
        pose_t far = shift_left(arc_end(ex[a].p, Ln, 0.0), W);
        entry.x = far.x; entry.y = far.y; entry.th = far.th + PI;
    }
    int route = 0;
    for (int a = 0; a < 4; ++a)
        for (int d = 0; d < 4; ++d) {
            const int m = ((d - 1 - a + 8) & 3) + 1;          /* arms whose ring arc the route drives: a .. a + m - 1 */
            road_t rd[COPO_MAX_SEGS];
            int n = 0;
            rd[n++] = in[a]; rd[n++] = bend[a]; rd[n++] = ring[a];
            for (int j = 1; j < m; ++j) { rd[n++] = join[(a + j - 1) & 3]; rd[n++] = ring[(a + j) & 3]; }
            rd[n++] = out[(a + m - 1) & 3]; rd[n++] = ex[(a + m - 1) & 3];
            emit_route(segs, meta, route++, rd, n);
        }
    *R = route;
    *P = emit_spawns(tab, sps, 4, 4);
}

/* name: "intersection" | "roundabout" (the maps' default parameters).  route_segs [16][COPO_MAX_SEGS + 1][COPO_SEG_STRIDE], route_meta
 * [16][4], spawn_tab [48][4], spawn_s [48].  Returns 0, or COPO_ERR_CONFIG for a map this file does not hold. */
int oracle_map_tables(const char* name, float* route_segs, float* route_meta, int32_t* spawn_tab, float* spawn_s,
                      int32_t* n_routes, int32_t* n_spawns) {
    if (!name || !route_segs || !route_meta || !spawn_tab || !spawn_s || !n_routes || !n_spawns) return COPO_ERR_NULL;
    if (!strcmp(name, "intersection")) { intersection(route_segs, route_meta, spawn_tab, spawn_s, n_routes, n_spawns); return COPO_OK; }
    if (!strcmp(name, "roundabout")) { roundabout(route_segs, route_meta, spawn_tab, spawn_s, n_routes, n_spawns); return COPO_OK; }
    return COPO_ERR_CONFIG;
}
