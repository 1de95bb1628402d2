"""Route-table maps for the vectorised simulator (build-defined scenes).

The reference gets its maps from MetaDrive's PG block library (`MultiAgent{Intersection,Roundabout,
Tollgate,ParkingLot}Env`, train_copo.py:1-2), whose source is not in the reference tree.  Here a map is
pure data consumed by `libcopo_hip.so`: a set of routes, each a start pose followed by (length,
curvature) pieces, and a set of spawn points on the first (straight) piece of the routes.

A vehicle's road is the corridor `[-lat_right, +lat_left]` around its route centreline, so maps with
different topology differ only in these tables -- the HIP kernel is map-agnostic.
"""
import math
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

MAX_SEGS = 8          # COPO_MAX_SEGS
SEG_STRIDE = 8        # COPO_SEG_STRIDE
MAX_ARC = math.radians(100.0)  # arcs are split so that the in-kernel atan2 never wraps
LANE_WIDTH = 3.5


@dataclass
class MapTables:
    name: str
    route_segs: np.ndarray   # [R][MAX_SEGS+1][8] f32: x0, y0, cos0, sin0, len, kappa, s_start, theta0
    route_meta: np.ndarray   # [R][4] f32: total_len, lat_left, lat_right, nseg
    spawn_tab: np.ndarray    # [P][4] i32: first_route, n_choices, 0, 0
    spawn_s: np.ndarray      # [P] f32
    default_num_agents: int
    extent: float = 100.0
    entries: List[Tuple[int, int]] = field(default_factory=list)

    @property
    def n_routes(self):
        return int(self.route_segs.shape[0])

    @property
    def n_spawns(self):
        return int(self.spawn_tab.shape[0])


def _wrap(a):
    return (a + math.pi) % (2 * math.pi) - math.pi


def build_route(x, y, th, pieces):
    """Integrate (length, kappa) pieces from pose (x, y, th) in float64; returns [MAX_SEGS+1][8] + length."""
    split = []
    for ln, kap in pieces:
        if ln <= 1e-9:
            continue
        n = 1 if kap == 0 else max(1, int(math.ceil(abs(kap) * ln / MAX_ARC)))
        split += [(ln / n, kap)] * n
    if len(split) > MAX_SEGS:
        raise ValueError("route needs %d segments > MAX_SEGS" % len(split))
    rec = np.zeros((MAX_SEGS + 1, SEG_STRIDE), np.float64)
    s = 0.0
    for k, (ln, kap) in enumerate(split):
        rec[k] = [x, y, math.cos(th), math.sin(th), ln, kap, s, _wrap(th)]
        if kap == 0:
            x, y = x + math.cos(th) * ln, y + math.sin(th) * ln
        else:
            r = 1.0 / kap
            x, y = x + r * (math.sin(th + kap * ln) - math.sin(th)), y - r * (math.cos(th + kap * ln) - math.cos(th))
            th = th + kap * ln
        s += ln
    for k in range(len(split), MAX_SEGS + 1):   # terminal record(s): end pose, zero length
        rec[k] = [x, y, math.cos(th), math.sin(th), 0.0, 0.0, s, _wrap(th)]
    return rec, s, len(split)


def _rot(x, y, th, q):
    c, s = math.cos(q), math.sin(q)
    return c * x - s * y, s * x + c * y, th + q


class _Builder:
    def __init__(self, name, default_num_agents, extent):
        self.name, self.n, self.extent = name, default_num_agents, extent
        self.routes, self.meta, self.spawn_tab, self.spawn_s, self.entries = [], [], [], [], []

    def add_entry(self, pose, route_pieces, lat_left, lat_right, spawn_offsets):
        """One entry lane: `route_pieces` is a list of piece-lists (one per destination)."""
        first = len(self.routes)
        seg0 = None
        for pieces in route_pieces:
            rec, total, nseg = build_route(*pose, pieces)
            if rec[0][5] != 0.0:
                raise ValueError("the first piece of a route must be straight (spawn pieces)")
            seg0 = rec[0][4] if seg0 is None else min(seg0, rec[0][4])
            self.routes.append(rec)
            self.meta.append([total, lat_left, lat_right, nseg])
        for s0 in spawn_offsets:
            if s0 >= seg0:
                raise ValueError("spawn offset %.1f beyond the first straight piece %.1f" % (s0, seg0))
            self.spawn_tab.append([first, len(route_pieces), 0, 0])
            self.spawn_s.append(s0)
        self.entries.append((first, len(route_pieces)))

    def finish(self):
        return MapTables(
            self.name, np.asarray(self.routes, np.float64).astype(np.float32),
            np.asarray(self.meta, np.float64).astype(np.float32), np.asarray(self.spawn_tab, np.int32),
            np.asarray(self.spawn_s, np.float32), self.n, self.extent, self.entries)


def intersection(exit_length=60.0, box=12.0, lane_width=LANE_WIDTH, spawns_per_lane=5, spawn_gap=9.0):
    """4-way, 2 lanes per direction.  Inner lane: left + straight; outer lane: straight + right."""
    b = _Builder("intersection", 30, box + exit_length)
    J, L, w = box, exit_length, lane_width
    offs = [2.0 + spawn_gap * k for k in range(spawns_per_lane)]
    for arm in range(4):
        q = arm * math.pi / 2
        for lane in range(2):
            a = w * (0.5 + lane)
            pose = _rot(-(J + L), -a, 0.0, q)
            straight = [(L, 0.0), (2 * J, 0.0), (L, 0.0)]
            left = [(L, 0.0), ((J + a) * math.pi / 2, 1.0 / (J + a)), (L, 0.0)]
            right = [(L, 0.0), ((J - a) * math.pi / 2, -1.0 / (J - a)), (L, 0.0)]
            if lane == 0:
                b.add_entry(pose, [left, straight], w * 0.5, w * 1.5, offs)
            else:
                b.add_entry(pose, [straight, right], w * 1.5, w * 0.5, offs)
    return b.finish()


def roundabout(exit_length=60.0, ring_radius=16.0, entry_radius=12.0, lane_width=LANE_WIDTH, spawns_per_lane=5,
               spawn_gap=9.0):
    """4-arm, 2-lane counter-clockwise ring.  Inner lane: straight + left; outer lane: right + straight."""
    w, re_ = lane_width, entry_radius
    geo = []
    for lane in range(2):
        a, R = w * (0.5 + lane), ring_radius + w * lane
        xc = -math.sqrt((R + re_) ** 2 - (a + re_) ** 2)
        geo.append((a, R, xc, math.asin(-xc / (R + re_))))
    x_start = -(exit_length + max(-g[2] for g in geo))
    b = _Builder("roundabout", 40, -x_start)
    offs = [2.0 + spawn_gap * k for k in range(spawns_per_lane)]
    for arm in range(4):
        q = arm * math.pi / 2
        for lane in range(2):
            a, R, xc, al = geo[lane]
            pose = _rot(x_start, -a, 0.0, q)
            lead = xc - x_start

            def route(k):
                return [(lead, 0.0), (al * re_, -1.0 / re_), ((2 * al + (k - 2) * math.pi / 2) * R, 1.0 / R),
                        (al * re_, -1.0 / re_), (lead, 0.0)]

            if lane == 0:
                b.add_entry(pose, [route(2), route(3)], w * 0.5, w * 1.5, offs)
            else:
                b.add_entry(pose, [route(1), route(2)], w * 1.5, w * 0.5, offs)
    return b.finish()


def tollgate(length=140.0, lanes=3, lane_width=LANE_WIDTH, spawns_per_lane=7, spawn_gap=9.0):
    """Two-direction straight road with `lanes` lanes each way (the gate itself is not modelled yet)."""
    b = _Builder("tollgate", 40, length / 2)
    w = lane_width
    offs = [2.0 + spawn_gap * k for k in range(spawns_per_lane)]
    for direction in range(2):
        q = direction * math.pi
        for lane in range(lanes):
            a = w * (0.5 + lane)
            pose = _rot(-length / 2, -a, 0.0, q)
            b.add_entry(pose, [[(length * 0.5, 0.0), (length * 0.5, 0.0)]], a, w * lanes - a, offs)
    return b.finish()


def _lane_change(shift, radius):
    """Two opposite arcs that move a lane centreline `shift` metres to the left (negative: to the right)."""
    if abs(shift) < 1e-9:
        return []
    phi = math.acos(1.0 - abs(shift) / (2.0 * radius))
    k = math.copysign(1.0 / radius, shift)
    return [(radius * phi, k), (radius * phi, -k)]


def bottleneck(approach=60.0, neck=30.0, lanes_wide=4, lanes_narrow=2, lane_width=LANE_WIDTH, taper_radius=40.0,
               spawns_per_lane=5, spawn_gap=9.0):
    """Two-direction road that narrows from `lanes_wide` to `lanes_narrow` lanes per direction and widens again
    (MetaDrive's MultiAgentBottleneckEnv, 20 agents: eval/evaluate_population.py:118-124).  Entry lane i merges into neck
    lane i * narrow // wide and leaves on one of the exit lanes that neck lane feeds; the corridor of a route is the
    neck's (the narrowest part)."""
    b = _Builder("bottleneck", 20, approach + neck + 40.0)
    w = lane_width
    per = lanes_wide // lanes_narrow
    assert per * lanes_narrow == lanes_wide, "lanes_wide must be a multiple of lanes_narrow"
    offs = [2.0 + spawn_gap * k for k in range(spawns_per_lane)]

    def run(shift):     # longitudinal length of a lane change
        return 2.0 * taper_radius * math.sin(math.acos(1.0 - abs(shift) / (2.0 * taper_radius)))

    # the widest lateral move fixes the taper zone: every route pads its straights so that all necks span the same x
    taper = max(run(w * (0.5 + lane) - w * (0.5 + lane // per)) for lane in range(lanes_wide))
    x0 = -(approach + taper + neck * 0.5)
    for direction in range(2):
        q = direction * math.pi
        for lane in range(lanes_wide):
            a_in = w * (0.5 + lane)
            j = lane // per
            a_neck = w * (0.5 + j)
            routes = []
            for k in range(j * per, (j + 1) * per):
                a_out = w * (0.5 + k)
                routes.append([(approach + taper - run(a_in - a_neck), 0.0)] + _lane_change(a_in - a_neck, taper_radius) +
                              [(neck, 0.0)] + _lane_change(a_neck - a_out, taper_radius) +
                              [(approach + taper - run(a_neck - a_out), 0.0)])
            b.add_entry(_rot(x0, -a_in, 0.0, q), routes, a_neck, w * lanes_narrow - a_neck, offs)
    return b.finish()


def parkinglot(spaces=8, aisle_half=40.0, lane_width=LANE_WIDTH, turn_radius=5.0, depth=6.0):
    """Aisle along x with `spaces` perpendicular parking spaces; agents leave spaces or drive into them."""
    b = _Builder("parkinglot", 10, aisle_half + 10.0)
    w, r = lane_width, turn_radius
    per_side = spaces // 2
    xs = [(-per_side / 2 + 0.5 + k) * 7.0 for k in range(per_side)]
    spots = [(x, +1) for x in xs] + [(x, -1) for x in xs]   # side +1: above the aisle, -1: below
    for x0, side in spots:                                   # out of a space, then east or west along the aisle
        th = -side * math.pi / 2                             # heading towards the aisle
        y_start = side * (w + depth + r)
        routes = []
        for go_east in (True, False):
            # lane centre y for travelling east is -w/2, for west +w/2
            y_lane = -w / 2 if go_east else w / 2
            turn_left = (side > 0) == go_east                # from above heading -y: east (+x) is a left turn
            lead = abs(y_start - y_lane) - r
            kap = (1.0 if turn_left else -1.0) / r
            x_after = x0 + (r if go_east else -r)
            run = (aisle_half - x_after) if go_east else (x_after + aisle_half)
            routes.append([(lead, 0.0), (r * math.pi / 2, kap), (run, 0.0)])
        b.add_entry((x0, y_start, th), routes, w * 0.5, w * 0.5, [0.5])
    for go_east in (True, False):                            # from an entrance into one of the spaces
        q = 0.0 if go_east else math.pi
        y_lane = -w / 2 if go_east else w / 2
        routes = []
        for x0, side in spots:
            turn_left = (side > 0) == go_east
            x_turn = x0 - r if go_east else x0 + r
            lead = (x_turn + aisle_half) if go_east else (aisle_half - x_turn)
            y_end = side * (w + depth + r)
            tail = abs(y_end - y_lane) - r
            routes.append([(lead, 0.0), (r * math.pi / 2, (1.0 if turn_left else -1.0) / r), (tail, 0.0)])
        x_s = -aisle_half if go_east else aisle_half
        b.add_entry((x_s, y_lane, q), routes, w * 0.5, w * 0.5, [1.0, 8.0, 15.0])
    return b.finish()


def pgmap(sequence="SCS", seed=0, lanes=2, lane_width=LANE_WIDTH, lead=50.0, spawns_per_lane=5, spawn_gap=9.0):
    """Two-way road assembled from a block sequence, in the manner of MetaDrive's procedurally generated maps (the
    `MultiAgentMetaDrive` env of train_all_copo_dist.py:10,30): `S` = straight of 40-80 m, `C` = curve of radius 30-60 m
    through 30-90 degrees to a random side; an int `sequence` draws that many blocks.  Everything random comes from
    `seed`, so a (sequence, seed) pair names one map.  `lead` metres of straight road at both ends hold the spawn points;
    vehicles enter at one end and leave at the other.  Junction blocks (ramps, roundabouts, intersections inside a chain)
    are not generated."""
    rng = np.random.RandomState(int(seed))
    if isinstance(sequence, (int, np.integer)):
        sequence = "".join("SC"[int(rng.randint(2))] for _ in range(int(sequence)))
    centre, heading = [], 0.0
    for ch in str(sequence):
        if ch == "S":
            centre.append((float(rng.uniform(40.0, 80.0)), 0.0))
        elif ch == "C":
            radius, ang = float(rng.uniform(30.0, 60.0)), math.radians(float(rng.uniform(30.0, 90.0)))
            side = 1.0 if rng.rand() < 0.5 else -1.0
            if abs(heading + side * ang) > math.radians(120.0):     # keep the chain from folding back over itself
                side = -side
            heading += side * ang
            centre.append((radius * ang, side / radius))
        else:
            raise ValueError("pgmap block %r: only S (straight) and C (curve) are generated" % ch)
    if len(centre) + 2 > MAX_SEGS:
        raise ValueError("pgmap: %d blocks + 2 lead pieces > %d route segments" % (len(centre), MAX_SEGS))
    chain = [(lead, 0.0)] + centre + [(lead, 0.0)]
    rec, total, nseg = build_route(0.0, 0.0, 0.0, chain)
    xe, ye, the = float(rec[nseg][0]), float(rec[nseg][1]), float(rec[nseg][7])
    # shift the road so that the midpoint of its two ends is the origin (scenes are centred like the other maps)
    ox, oy = -0.5 * xe, -0.5 * ye
    b = _Builder("pgmap", 20, 0.5 * total)
    w = lane_width
    offs = [2.0 + spawn_gap * k for k in range(spawns_per_lane)]
    if offs[-1] >= lead:
        raise ValueError("pgmap: spawn points do not fit on the %.0f m lead" % lead)

    def offset_right(pieces, a):       # the lane `a` metres to the right of a centre line
        return [(ln * (1.0 + k * a), k / (1.0 + k * a)) for ln, k in pieces]

    for direction in range(2):
        if direction == 0:
            x, y, th, pieces = ox, oy, 0.0, chain
        else:
            x, y, th, pieces = xe + ox, ye + oy, the + math.pi, [(ln, -k) for ln, k in reversed(chain)]
        for lane in range(lanes):
            a = w * (0.5 + lane)
            pose = (x + a * math.sin(th), y - a * math.cos(th), th)
            b.add_entry(pose, [offset_right(pieces, a)], a, w * lanes - a, offs)
    return b.finish()


MAP_BUILDERS = dict(intersection=intersection, roundabout=roundabout, tollgate=tollgate, parkinglot=parkinglot,
                    bottleneck=bottleneck, pgmap=pgmap)


def bounding_box(tables: MapTables, step=2.0):
    """(x_min, x_max, y_min, y_max) of the road network: route centrelines widened by their corridor.  Stands in for
    MetaDrive's `road_network.get_bounding_box()` (env_wrappers.py:268), used by the traffic-light columns."""
    lo, hi = np.array([np.inf, np.inf]), np.array([-np.inf, -np.inf])
    for r in range(tables.n_routes):
        pts = route_points(tables, r, step)
        pad = float(max(tables.route_meta[r, 1], tables.route_meta[r, 2]))
        lo, hi = np.minimum(lo, pts.min(0) - pad), np.maximum(hi, pts.max(0) + pad)
    return float(lo[0]), float(hi[0]), float(lo[1]), float(hi[1])


def ray_table(num_lasers):
    ang = 2.0 * np.pi * np.arange(num_lasers, dtype=np.float64) / num_lasers
    return np.stack([np.cos(ang), np.sin(ang)], axis=1).astype(np.float32)


def route_points(tables: MapTables, route: int, step=1.0):
    """Polyline of a route centreline (float64, for tests/plots)."""
    pts = []
    nseg = int(tables.route_meta[route, 3])
    for k in range(nseg):
        x0, y0, c0, s0, ln, kap, _, th0 = tables.route_segs[route, k].astype(np.float64)
        for s in np.arange(0.0, ln, step):
            if kap == 0:
                pts.append((x0 + c0 * s, y0 + s0 * s))
            else:
                r = 1.0 / kap
                pts.append((x0 + r * (math.sin(th0 + kap * s) - math.sin(th0)),
                            y0 - r * (math.cos(th0 + kap * s) - math.cos(th0))))
    pts.append(tuple(tables.route_segs[route, nseg, :2].astype(np.float64)))
    return np.asarray(pts)
