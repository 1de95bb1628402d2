"""Road-network maps for the vectorised simulator.

The reference gets its maps from MetaDrive 0.2.5's PG block library (`MultiAgent{Intersection,Roundabout,Tollgate,
Bottleneck,ParkingLot}Env`, train_copo.py:1-2), whose source is not in the reference tree.  What is rebuilt here is the
*structure* those environments hand to the observation / reward code, in the form the kernels consume:

* a map is a directed graph of ROADS; a road is one geometric primitive (a straight or a circular arc, like MetaDrive's
  StraightLane / CircularLane) carrying `lanes` parallel lanes.  The primitive stored is the centre line of lane 0 (the
  leftmost lane); lane i lies `i * lane_width` to its right (concentric for arcs);
* vehicles spawn in slots on the SPAWN ROADS and are sent to the far end of the reverse of a random spawn road
  (MetaDrive's `_update_destination_for`: `-choice(spawn_roads)`), over the breadth-first shortest road sequence;
* a ROUTE = that road sequence.  `route_segs[r][k]` is one road of route r (COPO_SEG_STRIDE floats, see `SEG_*`),
  including what the navigation observation needs per road (the check point at the road's end, the three curve features).

Geometry pinned by self-consistency: the Intersection (turn radius 10, two lanes: right turns of radius 10 / 13.5, left
turns 17 / 20.5, crossing 30.5 m, U-turn 1.75 / 5.25) and the Roundabout (exit radius 10, inner radius 30, angle 70 deg)
follow the published block formulas; `tests/test_abi_and_sim_cpu.py` checks that the roundabout's ring closes to < 1 mm
when every arm is built from its own entry, which only happens for the right connecting radius.
Coordinates are right-handed (x east, y north, headings counter-clockwise); "left" is +lateral.
"""
import math
from collections import OrderedDict, deque
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

MAX_SEGS = 16         # COPO_MAX_SEGS: roads per route (a full turn of the roundabout is 11)
SEG_STRIDE = 16       # COPO_SEG_STRIDE
LANE_WIDTH = 3.5
# columns of a segment (road) record
(SEG_X0, SEG_Y0, SEG_COS, SEG_SIN, SEG_LEN, SEG_KAPPA, SEG_S0, SEG_TH0, SEG_CKX, SEG_CKY, SEG_LANES, SEG_F_RADIUS,
 SEG_RADIUS, SEG_F_ANGLE, SEG_UMX, SEG_UMY) = range(16)

NAVI_RADIUS_NORM = 60.0   # MetaDrive BlockParameterSpace.CURVE radius max; the feature divides by (this + lanes * width)
NAVI_ANGLE_NORM = 135.0   # ... CURVE angle max (degrees)
RESPAWN_REGION_LONGITUDE = 8.0   # MetaDrive SpawnManager: slot pitch and the box that must be free for a respawn
RESPAWN_REGION_LATERAL = 3.0
ENTRANCE_LENGTH = 10.0           # FirstPGBlock: the first 10 m of the first block are not a spawn road


@dataclass
class MapTables:
    name: str
    route_segs: np.ndarray   # [R][MAX_SEGS+1][SEG_STRIDE] f32
    route_meta: np.ndarray   # [R][4] f32: total_len (lane-0 line), nseg, index of the toll-booth road (-1: none), exclusive destination id + 1 (0: none)
    spawn_tab: np.ndarray    # [P][4] i32: first_route, n_destinations, lane, safe (1 = a respawn place)
    spawn_s: np.ndarray      # [P] f32 longitudinal position of the slot on the spawn road
    default_num_agents: int
    lane_width: float = LANE_WIDTH
    extent: float = 100.0
    entries: List[Tuple[int, int]] = field(default_factory=list)
    lines: np.ndarray = None  # [n][8] f32 lane-line primitives for the side / lane-line detectors (see `Net.lines`)
    boxes: np.ndarray = None  # [n][6] f32 static boxes (buildings): centre x, y, cos, sin of the long axis, half length, half width

    @property
    def n_routes(self):
        return int(self.route_segs.shape[0])

    @property
    def n_spawns(self):
        return int(self.spawn_tab.shape[0])


def _wrap(a):
    return (a + math.pi) % (2 * math.pi) - math.pi


def advance(pose, length, kappa):
    """Pose after `length` metres of constant curvature `kappa` (float64)."""
    x, y, th = pose
    if kappa == 0:
        return x + math.cos(th) * length, y + math.sin(th) * length, th
    r = 1.0 / kappa
    return (x + r * (math.sin(th + kappa * length) - math.sin(th)),
            y - r * (math.cos(th + kappa * length) - math.cos(th)), th + kappa * length)


def shift(pose, left):
    """Pose moved `left` metres to its left (negative: to the right)."""
    x, y, th = pose
    return x - math.sin(th) * left, y + math.cos(th) * left, th


def reverse(pose):
    return pose[0], pose[1], pose[2] + math.pi


# lane-line kinds for the detectors (MetaDrive: the side detector sees continuous lines, the lane-line detector both)
LINE_BROKEN, LINE_CONTINUOUS = 1.0, 2.0


class Net:
    """Directed road graph.  A road (a, b) = (lane-0 start pose, length, kappa of the lane-0 line, lanes)."""

    def __init__(self, lane_width=LANE_WIDTH):
        self.w = lane_width
        self.roads = OrderedDict()
        self.adj = OrderedDict()
        self.lines = []          # [x0, y0, theta0, length, kappa, kind, 0, 0]
        self.toll_roads = set()  # roads that are toll booths (Tollgate)
        self.solid = {}          # road -> (left edge continuous, right edge continuous)
        self.funnel = {}         # road -> (wave radius, extra width at the wide end, +1 narrowing / -1 widening): Merge / Split blocks
        self.open_left = set()   # roads whose broken left edge line may be crossed (one lane width of the opposite direction)
        self.open_left_if_broken = False

    def add(self, a, b, pose, length, kappa, lanes, left_line=LINE_CONTINUOUS, right_line=LINE_CONTINUOUS,
            inner_line=LINE_BROKEN, toll=False, open_left=False):
        """`pose` is the start of lane 0's centre line.  Line kinds: 0 = none (inside junctions)."""
        assert (a, b) not in self.roads, (a, b)
        self.roads[(a, b)] = (tuple(float(v) for v in pose), float(length), float(kappa), int(lanes))
        # edge lines that a vehicle's body must not touch (MetaDrive: on_yellow / on_white_continuous_line): continuous ones
        self.solid[(a, b)] = (left_line == LINE_CONTINUOUS, right_line == LINE_CONTINUOUS)
        # a BROKEN left edge line with the opposite direction's lane behind it may be crossed (MetaDrive: the vehicle is still
        # `on_lane`; only continuous lines and the sidewalk end an agent): maps that model it set `open_left_if_broken`
        if getattr(self, "open_left_if_broken", False) and (left_line == LINE_BROKEN or open_left):
            self.open_left.add((a, b))
        if toll:
            self.toll_roads.add((a, b))
        self.adj.setdefault(a, []).append(b)
        self.adj.setdefault(b, [])
        w = self.w
        for i in range(lanes + 1):   # boundary i lies (i - 0.5) * w to the right of lane 0
            kind = left_line if i == 0 else (right_line if i == lanes else inner_line)
            if kind:
                a_off = -(i - 0.5) * w
                p = shift(pose, a_off)
                k = kappa / (1.0 - kappa * a_off) if kappa else 0.0
                ln = length * (1.0 - kappa * a_off) if kappa else length
                self.lines.append([p[0], p[1], p[2], ln, k, kind, 0.0, 0.0])
        return advance(pose, length, kappa)

    def add_funnel(self, a, b, pose, length, lanes, extra_lanes, narrowing, centre=LINE_CONTINUOUS):
        """MetaDrive's Merge ("y", `narrowing`) / Split ("Y") block: the route follows a straight `lanes`-lane road of
        `length` (Bottleneck.BOTTLENECK_LEN = 20 m); `extra_lanes` more lanes to its right run into it / out of it on WAVE lanes
        -- two arcs of opposite sense, `create_wave_lanes`: lane `index` is shifted by `index * lane_width` over the length, half
        of it by each arc, angle = pi - 2 atan(length / (2 * lateral_dist)), radius = length / (2 sin(angle)) with lateral_dist
        = index * lane_width / 2.  Those lanes are drivable but not part of the route: the road record keeps `lanes` (check
        points, lane index) and carries the outermost wave lane as extra width on its right (R, D, direction); the only lines
        are the centre line (continuous) and the outer edge of the outermost wave lane (continuous): two arcs."""
        w = self.w
        end = self.add(a, b, pose, length, 0.0, lanes, centre, 0, 0)
        self.solid[(a, b)] = (centre == LINE_CONTINUOUS, True)
        d = extra_lanes * w / 2.0                              # lateral_dist of the outermost wave lane, per arc
        ang = math.pi - 2.0 * math.atan(length / (2.0 * d))
        if not 0.0 < ang < math.pi / 2:      # (each arc turns by less than a quarter: the edge arcs R +- w / 2 stay real for the width function)
            raise ValueError("funnel of %d extra lanes over %.1f m: the wave lanes would turn by %.0f degrees" % (extra_lanes, length, math.degrees(ang)))
        R = length / (2.0 * math.sin(ang))
        # record fields of the funnel: wave radius, extra width at the wide end (+ narrowing / - widening), and how far from the
        # wide end the edge line's first arc (radius R + w / 2: the edge runs outside that bend) hands over to the second (R - w / 2)
        self.funnel[(a, b)] = (R, (2.0 * d) if narrowing else -(2.0 * d), (R + 0.5 * w) * math.sin(ang))
        # outer edge line = the outermost wave lane's centre line shifted w / 2 to the right.  Narrowing: it starts (lanes +
        # extra - 0.5) w to the right of lane 0 and bends LEFT first, then right; widening: it starts (lanes - 0.5) w to the
        # right and bends RIGHT first, then left.
        first = 1.0 if narrowing else -1.0                      # sense of the first arc (+ = left)
        off0 = (lanes + (extra_lanes if narrowing else 0) - 1) * w      # outermost wave lane's centre at the start, to the right of lane 0
        c = shift(pose, -off0)
        for sense in (first, -first):
            kap = sense / R
            a_off = -0.5 * w                                    # the lane's right edge
            p = shift(c, a_off)
            k = kap / (1.0 - kap * a_off)
            self.lines.append([p[0], p[1], p[2], R * ang * (1.0 - kap * a_off), k, LINE_CONTINUOUS, 0.0, 0.0])
            c = advance(c, R * ang, kap)
        return end

    def end_pose(self, a, b):
        pose, ln, k, _ = self.roads[(a, b)]
        return advance(pose, ln, k)

    def adverse(self, a, b, na, nb, **kw):
        """CreateAdverseRoad: the opposite carriageway of road (a, b) as road (na, nb): same lane count, lane 0 next to
        lane 0, i.e. the lane-0 lines are one lane width apart."""
        pose, ln, k, lanes = self.roads[(a, b)]
        end = advance(pose, ln, k)
        start = reverse(shift(end, self.w))
        kr = -k / (1.0 - k * self.w) if k else 0.0
        lr = ln * (1.0 - k * self.w) if k else ln
        return self.add(na, nb, start, lr, kr, lanes, **kw)

    def bfs(self, src, dst):
        """Node list of the shortest path by number of roads (ties: insertion order, as a queue-based search gives)."""
        prev, q = {src: None}, deque([src])
        while q:
            u = q.popleft()
            if u == dst:
                break
            for v in self.adj.get(u, []):
                if v not in prev:
                    prev[v] = u
                    q.append(v)
        if dst not in prev:
            raise ValueError("no route %s -> %s" % (src, dst))
        path = [dst]
        while prev[path[-1]] is not None:
            path.append(prev[path[-1]])
        return path[::-1]


def road_record(pose, length, kappa, lanes, s_start, w, solid=(False, False), funnel=None, open_left=False):
    """One SEG_STRIDE record (float64) for a road whose lane-0 line starts at `pose`.  The lanes field carries the edge-line
    flags in its fraction: lanes + 0.25 (left edge continuous) + 0.5 (right edge continuous) + 0.125 (left edge broken with a
    drivable lane of the opposite direction behind it: the centre may go one lane width beyond the line)."""
    x, y, th = pose
    rec = np.zeros(SEG_STRIDE, np.float64)
    rec[[SEG_X0, SEG_Y0, SEG_COS, SEG_SIN, SEG_LEN, SEG_KAPPA, SEG_S0, SEG_TH0]] = [
        x, y, math.cos(th), math.sin(th), length, kappa, s_start, _wrap(th)]
    end = advance(pose, length, kappa)
    ck = shift(end, -(lanes / 2.0 - 0.5) * w)          # end of the road, lateral middle (Navigation check point)
    rec[SEG_CKX], rec[SEG_CKY], rec[SEG_LANES] = ck[0], ck[1], lanes + 0.25 * bool(solid[0]) + 0.5 * bool(solid[1]) + 0.125 * bool(open_left and not solid[0])
    if kappa == 0:
        rec[SEG_F_RADIUS], rec[SEG_RADIUS], rec[SEG_F_ANGLE] = 0.0, 0.0, 0.5
        rec[SEG_UMX], rec[SEG_UMY] = 1.0, 0.0
        if funnel is not None:      # straight road of a Merge / Split block: wave radius, extra width at the wide end, direction
            rec[SEG_RADIUS], rec[SEG_UMX], rec[SEG_UMY] = funnel
    else:
        radius, ang = 1.0 / abs(kappa), abs(kappa) * length
        rec[SEG_F_RADIUS] = min(1.0, radius / (NAVI_RADIUS_NORM + lanes * w))
        rec[SEG_RADIUS] = radius                        # (the direction feature is the sign of kappa: + = left)
        rec[SEG_F_ANGLE] = min(1.0, (math.degrees(ang) / NAVI_ANGLE_NORM + 1.0) / 2.0)
        sg = 1.0 if kappa > 0 else -1.0
        # unit vector from the arc's centre to its mid point: the start radial -sg*n0 turned by sg*ang/2
        ux, uy = sg * math.sin(th), -sg * math.cos(th)
        c, s = math.cos(sg * ang / 2.0), math.sin(sg * ang / 2.0)
        rec[SEG_UMX], rec[SEG_UMY] = c * ux - s * uy, s * ux + c * uy
    return rec


class _Builder:
    def __init__(self, name, net, default_num_agents, extent):
        self.name, self.net, self.n, self.extent = name, net, default_num_agents, extent
        self.routes, self.meta, self.spawn_tab, self.spawn_s, self.entries = [], [], [], [], []

    def add_route(self, nodes):
        net, w = self.net, self.net.w
        if len(nodes) - 1 > MAX_SEGS:
            raise ValueError("route needs %d roads > MAX_SEGS" % (len(nodes) - 1))
        rec = np.zeros((MAX_SEGS + 1, SEG_STRIDE), np.float64)
        s, lanes, toll, solid = 0.0, 1, -1, (False, False)
        for k in range(len(nodes) - 1):
            pose, ln, kap, lanes = net.roads[(nodes[k], nodes[k + 1])]
            solid = net.solid[(nodes[k], nodes[k + 1])]
            if (nodes[k], nodes[k + 1]) in net.toll_roads:
                toll = k
            rec[k] = road_record(pose, ln, kap, lanes, s, w, solid, net.funnel.get((nodes[k], nodes[k + 1])),
                                 open_left=(nodes[k], nodes[k + 1]) in net.open_left)
            s += ln
        end = net.end_pose(nodes[-2], nodes[-1])
        nseg = len(nodes) - 1
        for k in range(nseg, MAX_SEGS + 1):     # terminal record(s): end pose, zero length
            rec[k] = road_record(end, 0.0, 0.0, lanes, s, w, solid)
        self.routes.append(rec)
        self.meta.append([s, nseg, toll, 0.0])

    def add_spawn_road(self, road, destinations, slot_longs, safe_only_first=True, lanes=None, exclusive=None):
        """Routes from `road` (a, b) to every destination node; slots on each of its lanes at `slot_longs`.
        `exclusive`: per destination an id >= 0 of a place that only ONE living agent may be heading for at a time (MetaDrive's
        ParkingSpaceManager: a parking space is handed out once and comes back when its agent is done) -> route_meta[r][3] = id + 1."""
        net = self.net
        first = len(self.routes)
        for k, d in enumerate(destinations):
            nodes = [road[0]] + net.bfs(road[1], d)
            self.add_route(nodes)
            if exclusive is not None:
                self.meta[-1][3] = float(exclusive[k] + 1)
        n_l = net.roads[road][3] if lanes is None else lanes
        if slot_longs and max(slot_longs) >= net.roads[road][1]:
            raise ValueError("spawn slot beyond the spawn road")
        for lane in range(n_l):
            for j, s0 in enumerate(slot_longs):
                self.spawn_tab.append([first, len(destinations), lane, 1 if (j == 0 or not safe_only_first) else 0])
                self.spawn_s.append(s0)
        self.entries.append((first, len(destinations)))

    def finish(self):
        return MapTables(
            self.name, np.asarray(self.routes, np.float64).astype(np.float32),
            np.asarray(self.meta, np.float64).astype(np.float32), np.asarray(self.spawn_tab, np.int32),
            np.asarray(self.spawn_s, np.float32), self.n, self.net.w, self.extent, self.entries,
            np.asarray(self.net.lines, np.float64).astype(np.float32).reshape(-1, 8))


def spawn_slots(exit_length):
    """SpawnManager._auto_fill_spawn_roads_randomly: floor(L / 8) slots, pitch L / slots, the first at 4 m."""
    eff = exit_length - ENTRANCE_LENGTH
    n = int(math.floor(eff / RESPAWN_REGION_LONGITUDE))
    return [RESPAWN_REGION_LONGITUDE / 2 + j * (eff / n) for j in range(n)]


def intersection(exit_length=60.0, radius=10.0, lanes=2, lane_width=LANE_WIDTH, u_turn=True):
    """MAIntersectionMap: FirstPGBlock + InterSection(radius 10), `lanes` lanes per direction, every entry lane may turn
    left / go straight / turn right, U-turns at the stop line when lanes > 1.  Arm 0 is the first block (entry road
    50 m = exit_length - 10, its exit road as well), the other arms are 60 m."""
    w, n = lane_width, lanes
    net = Net(w)
    half = radius + (2 * n - 1) * w / 2.0                # stop lines are this far from the junction centre
    cx, cy = exit_length + half, w / 2.0                 # MetaDrive's frame: entry lane 0 of arm 0 runs along y = 0
    r_right0 = radius + (n - 1) * w                      # lane 0 of a right turn (the rightmost lane has `radius`)
    r_left0 = radius + n * w
    arm_len = [exit_length - ENTRANCE_LENGTH] + [exit_length] * 3
    for a in range(4):
        psi = math.pi + a * math.pi / 2                  # outward direction of arm a (0 west, 1 south, 2 east, 3 north)
        h = psi + math.pi
        stop = (cx + half * math.cos(psi), cy + half * math.sin(psi), h)           # on the arm's axis, heading in
        entry0 = shift(advance(stop, -arm_len[a], 0.0), -w / 2.0)
        net.add("in%d" % a, "stop%d" % a, entry0, arm_len[a], 0.0, n)
        exit0 = shift((stop[0], stop[1], psi), -w / 2.0)
        net.add("out%d" % a, "end%d" % a, exit0, arm_len[a], 0.0, n)
    for a in range(4):
        e = net.end_pose("in%d" % a, "stop%d" % a)
        net.add("stop%d" % a, "out%d" % ((a + 1) % 4), e, r_right0 * math.pi / 2, -1.0 / r_right0, n, 0, LINE_CONTINUOUS, 0)
        net.add("stop%d" % a, "out%d" % ((a + 2) % 4), e, 2 * half, 0.0, n, 0, 0, 0)
        net.add("stop%d" % a, "out%d" % ((a + 3) % 4), e, r_left0 * math.pi / 2, 1.0 / r_left0, n, 0, 0, 0)
        if u_turn and n > 1:
            net.add("stop%d" % a, "out%d" % a, e, (w / 2) * math.pi, 2.0 / w, n, 0, 0, 0)
    b = _Builder("intersection", net, 30, half + exit_length)
    slots = spawn_slots(exit_length)
    dests = [d for d in range(4)]
    for a in range(4):
        ds = ["end%d" % d for d in dests if (d != a or (u_turn and n > 1))]
        b.add_spawn_road(("in%d" % a, "stop%d" % a), ds, slots)
    return b.finish()


def roundabout(exit_length=60.0, exit_radius=10.0, inner_radius=30.0, angle_deg=70.0, lanes=2, lane_width=LANE_WIDTH):
    """MARoundaboutMap: FirstPGBlock + Roundabout(exit radius 10, inner radius 30, angle 70).  Per arm: entry bend (right,
    `exit_radius` on the rightmost lane, `angle`), ring arc (left, radius_big, 2*angle - 90), exit bend (right), exit
    straight; consecutive arms are joined by a ring arc of 180 - 2*angle whose radius `beneath / cos(angle) - exit_radius`
    is the one that closes the ring."""
    w, n = lane_width, lanes
    net = Net(w)
    ang = math.radians(angle_deg)
    r_big = (2 * n - 1) * w + inner_radius                 # rightmost lane of the ring arcs
    beneath = (2 * n - 1) * w / 2.0 + exit_radius
    r_join = beneath / math.cos(ang) - exit_radius
    lane0 = lambda r_rightmost, left_turn: (r_rightmost - (n - 1) * w) if left_turn else (r_rightmost + (n - 1) * w)
    r_e0, r_b0, r_j0 = lane0(exit_radius, False), lane0(r_big, True), lane0(r_join, True)
    arm_len = [exit_length - ENTRANCE_LENGTH] + [exit_length] * 3
    entry0 = (ENTRANCE_LENGTH, 0.0, 0.0)                   # lane 0 of the first block's spawn road
    for a in range(4):
        e = net.add("in%d" % a, "stop%d" % a, entry0, arm_len[a], 0.0, n)
        p = net.add("stop%d" % a, "r%da" % a, e, r_e0 * ang, -1.0 / r_e0, n, LINE_BROKEN, LINE_CONTINUOUS)
        p1 = net.add("r%da" % a, "r%db" % a, p, r_b0 * (2 * ang - math.pi / 2), 1.0 / r_b0, n, LINE_CONTINUOUS, LINE_BROKEN)
        p2 = net.add("r%db" % a, "out%d" % a, p1, r_e0 * ang, -1.0 / r_e0, n, LINE_BROKEN, LINE_CONTINUOUS)
        nxt = (a + 1) % 4
        net.add("out%d" % a, "end%d" % nxt, p2, arm_len[nxt], 0.0, n)
        net.add("r%db" % a, "r%da" % nxt, p1, r_j0 * (math.pi - 2 * ang), 1.0 / r_j0, n, LINE_CONTINUOUS, LINE_BROKEN)
        # the next arm's entry is the adverse carriageway of this exit (lane 0 one lane width to the left, reversed)
        entry0 = reverse(shift(advance(p2, arm_len[nxt], 0.0), w))
    b = _Builder("roundabout", net, 40, 120.0)
    slots = spawn_slots(exit_length)
    for a in range(4):
        # arm a's traffic leaves through "end<d>"; end<a> (its own arm) means a full turn of the ring
        b.add_spawn_road(("in%d" % a, "stop%d" % a), ["end%d" % d for d in range(4)], slots)
    return b.finish()


def bottleneck(exit_length=60.0, bottle_lanes=4, neck_lanes=1, neck_length=20.0, taper=20.0, lane_width=LANE_WIDTH, centre_open=False):
    """MABottleneckMap (20 agents, eval/evaluate_population.py:118-124; `bottle_lane_num=4, neck_lane_num=1, neck_length=20`):
    FirstPGBlock -> Merge -> Split.  The first block's spawn road is `exit_length - 10` long (NODE_2 -> NODE_3), the Split's socket
    road `exit_length`; the Merge / Split blocks are funnels (`Net.add_funnel`): the route follows the `neck_lanes` leftmost lanes
    straight through, the other lanes bend into / out of them on wave lanes over `taper` = BOTTLENECK_LEN = 20 m."""
    w = lane_width
    net = Net(w)
    # centre_open (experiment, profiles/r06_fidelity.txt): the centre line of the Merge / neck / Split roads BROKEN and crossable by one lane
    # width (MetaDrive's BOTTLENECK_PARAMETER has a `solid_center_line` entry whose default the map does not override -- as remembered, unverified)
    net.open_left_if_broken = bool(centre_open)
    cl = LINE_BROKEN if centre_open else LINE_CONTINUOUS
    first = exit_length - ENTRANCE_LENGTH
    total = first + exit_length + 2 * taper + neck_length
    extra = bottle_lanes - neck_lanes
    for d in range(2):
        o = (0.0, 0.0, 0.0) if d == 0 else reverse(shift((total, 0.0, 0.0), w))
        l_in, l_out = (first, exit_length) if d == 0 else (exit_length, first)
        e = net.add("in%d" % d, "w%d" % d, o, l_in, 0.0, bottle_lanes)
        e = net.add_funnel("w%d" % d, "n%d" % d, e, taper, neck_lanes, extra, True, centre=cl)
        e = net.add("n%d" % d, "m%d" % d, e, neck_length, 0.0, neck_lanes, left_line=cl)
        e = net.add_funnel("m%d" % d, "x%d" % d, e, taper, neck_lanes, extra, False, centre=cl)
        net.add("x%d" % d, "end%d" % d, e, l_out, 0.0, bottle_lanes)
    b = _Builder("bottleneck", net, 20, total / 2)
    slots = spawn_slots(exit_length)
    for d in range(2):
        b.add_spawn_road(("in%d" % d, "w%d" % d), ["end%d" % d], slots)
    return b.finish()


def tollgate(exit_length=70.0, lanes=3, toll_lanes=8, toll_length=10.0, taper=20.0, socket=2.0, lane_width=LANE_WIDTH):
    """MATollGateMap (40 agents): FirstPGBlock (`lanes` lanes) -> Split (funnel to `toll_lanes` lanes over BOTTLENECK_LEN =
    `taper` = 20 m, socket road of 2 m) -> TollGate block (`toll_length`) -> Merge (funnel back, socket road = `exit_length`).
    The route follows the `lanes` leftmost lanes straight through both funnels (`Net.add_funnel`); the booth logic (a vehicle
    must spend `min_pass_steps` inside a booth) is the simulator's `toll` option."""
    w = lane_width
    net = Net(w)
    first = exit_length - ENTRANCE_LENGTH
    total = first + exit_length + 2 * taper + socket + toll_length
    extra = toll_lanes - lanes
    for d in range(2):
        o = (0.0, 0.0, 0.0) if d == 0 else reverse(shift((total, 0.0, 0.0), w))
        l_in, l_out = (first, exit_length) if d == 0 else (exit_length, first)
        e = net.add("in%d" % d, "f%d" % d, o, l_in, 0.0, lanes)
        e = net.add_funnel("f%d" % d, "s%d" % d, e, taper, lanes, extra, False)
        if d == 0:      # (the blocks are laid out along direction 0: its socket road precedes the booths, direction 1 meets it after them)
            e = net.add("s%d" % d, "t%d" % d, e, socket, 0.0, toll_lanes, LINE_CONTINUOUS, LINE_CONTINUOUS, 0)
            e = net.add("t%d" % d, "g%d" % d, e, toll_length, 0.0, toll_lanes, LINE_CONTINUOUS, LINE_CONTINUOUS, LINE_CONTINUOUS, toll=True)
        else:
            e = net.add("s%d" % d, "t%d" % d, e, toll_length, 0.0, toll_lanes, LINE_CONTINUOUS, LINE_CONTINUOUS, LINE_CONTINUOUS, toll=True)
            e = net.add("t%d" % d, "g%d" % d, e, socket, 0.0, toll_lanes, LINE_CONTINUOUS, LINE_CONTINUOUS, 0)
        e = net.add_funnel("g%d" % d, "x%d" % d, e, taper, lanes, extra, True)
        net.add("x%d" % d, "end%d" % d, e, l_out, 0.0, lanes)
    b = _Builder("tollgate", net, 40, total / 2)
    slots = spawn_slots(exit_length)
    for d in range(2):
        b.add_spawn_road(("in%d" % d, "f%d" % d), ["end%d" % d], slots)
    t = b.finish()
    # TollGate._add_building_and_speed_limit: `if idx % 2 == 1` a TollGateBuilding (the lane's width, the road's length) at the centre
    # of every second lane of the booth road, in both directions (used by the simulator when SimConfig.toll_buildings is on)
    boxes = []
    for d in range(2):
        pose, ln, _, n = net.roads[("t%d" % d, "g%d" % d) if d == 0 else ("s%d" % d, "t%d" % d)]
        for idx in range(1, n, 2):
            c = advance(shift(pose, -idx * w), ln / 2.0, 0.0)
            boxes.append([c[0], c[1], math.cos(c[2]), math.sin(c[2]), ln / 2.0, w / 2.0])
    t.boxes = np.asarray(boxes, np.float64).astype(np.float32)
    return t


def parkinglot(spaces=8, exit_length=20.0, lane_width=LANE_WIDTH, turn_radius=4.0, depth=8.0, arm=10.0, junction_radius=10.0,
               unique_spaces=True, centre_line_open=True):
    """MAParkinglotMap (10 agents, `parking_space_num = 8`, `exit_length = 20`, one lane per direction): FirstPGBlock ->
    ParkingLot block -> T-intersection (`t_type = 1`: the straight arm is missing, `EXIT_PART_LENGTH = 10`).
    ParkingLot block (`one_side_vehicle_num = spaces / 2`, `radius` 4, `length` 8): a main road of `2 r + (n - 1) w` = 18.5 m and a
    4 m socket road; `n` spaces of `w x 8` m side by side on either side, perpendicular to the road.  Every space is its own little
    road system, as MetaDrive builds it (`_add_one_parking_space`) -- overlapping roads, not cuts of one aisle:
      in from the near lane:  [straight `dist_to_in`] -> right bend (r) -> the space (8 m)
      in from the far lane:   [straight `dist_to_out`] -> left bend (r) -> straight `w` across the near lane -> the space
      out to the near lane:   the space reversed (its spawn road) -> right bend (r) -> [straight `dist_to_out`] -> next block
      out to the far lane:    the space reversed -> straight `w` -> left bend (r) -> [straight `dist_to_in`] -> previous block
    Entrances (spawn roads, one slot each at 4 m): the first block's 10 m spawn road and the two arms of the T; a vehicle that enters
    there is sent into a space, a vehicle that starts in a space (slot at 4 m of its 8 m) to the far end of one of the three exits.
    `unique_spaces` (MetaDrive's ParkingSpaceManager, envs/marl_envs/marl_parking_lot.py): an entrant is sent to a space that no
    other living agent is heading for; the space comes back when that agent is done.  `centre_line_open`: the broken centre line
    of the two-way roads may be crossed -- a vehicle in the opposite lane is still `on_lane` in MetaDrive; continuous lines and
    the sidewalk end an agent.  (Both False: the round-3 scene.)
    Frame: the positive lane runs along +x at y = 0, the negative lane back at y = w."""
    w, r = lane_width, turn_radius
    net = Net(w)
    net.open_left_if_broken = bool(centre_line_open)
    n = spaces // 2
    x0 = exit_length                                   # the ParkingLot block starts where the first block ends
    main = 2 * r + (n - 1) * w
    x1 = x0 + main                                     # socket road starts
    x2 = x1 + 4.0                                      # T-intersection starts (SOCKET_LENGTH = 4)
    NO = (0, 0, 0)
    # ---- first block (spawn road and its adverse = exit 1) ----------------------------------------------------------------
    net.add("in0", "P", (x0 - (exit_length - ENTRANCE_LENGTH), 0.0, 0.0), exit_length - ENTRANCE_LENGTH, 0.0, 1, LINE_BROKEN, LINE_CONTINUOUS)
    net.add("Q", "end0", (x0, w, math.pi), exit_length - ENTRANCE_LENGTH, 0.0, 1, LINE_BROKEN, LINE_CONTINUOUS)
    # ---- ParkingLot block: main road + socket, both directions ---------------------------------------------------------------
    net.add("P", "S", (x0, 0.0, 0.0), main, 0.0, 1, LINE_BROKEN, 0)
    net.add("S", "T", (x1, 0.0, 0.0), 4.0, 0.0, 1, LINE_BROKEN, LINE_CONTINUOUS)
    net.add("Tn", "Sn", (x2, w, math.pi), 4.0, 0.0, 1, LINE_BROKEN, LINE_CONTINUOUS)
    net.add("Sn", "Q", (x1, w, math.pi), main, 0.0, 1, LINE_BROKEN, 0)
    # ---- T-intersection (radius 10, one lane): right arm "A" (south), left arm "B" (north) ---------------------------------
    R, RL = junction_radius, junction_radius + w       # right / left turn radii of a one-lane junction
    cross = 2 * junction_radius + w                    # straight through: 2 r + (2 lanes - 1) w
    e = net.add("T", "Ax", (x2, 0.0, 0.0), R * math.pi / 2, -1.0 / R, 1, *NO)                       # parking road -> south arm
    net.add("Ax", "endA", e, arm, 0.0, 1, LINE_BROKEN, LINE_CONTINUOUS)
    e = net.add("T", "Bx", (x2, 0.0, 0.0), RL * math.pi / 2, 1.0 / RL, 1, *NO)                      # parking road -> north arm
    net.add("Bx", "endB", e, arm, 0.0, 1, LINE_BROKEN, LINE_CONTINUOUS)
    a_in = (x2 + R + w, -(R + arm), math.pi / 2)       # south arm's entrance lane: w to the left of its exit lane, heading north
    e = net.add("inA", "Ai", a_in, arm, 0.0, 1, LINE_BROKEN, LINE_CONTINUOUS)
    net.add("Ai", "Tn", e, RL * math.pi / 2, 1.0 / RL, 1, *NO)                                        # south arm -> parking road (left turn)
    net.add("Ai", "Bx", e, cross, 0.0, 1, *NO)                                                       # south arm -> north arm
    b_in = (x2 + R, w + R + arm, -math.pi / 2)         # north arm's entrance lane, heading south
    e = net.add("inB", "Bi", b_in, arm, 0.0, 1, LINE_BROKEN, LINE_CONTINUOUS)
    net.add("Bi", "Tn", e, R * math.pi / 2, -1.0 / R, 1, *NO)                                         # north arm -> parking road (right turn)
    net.add("Bi", "Ax", e, cross, 0.0, 1, *NO)                                                       # north arm -> south arm
    # ---- the spaces ------------------------------------------------------------------------------------------------------------
    dests = []
    for side in (-1, +1):                              # -1: south of the positive lane (MetaDrive's part 2), +1: north of the negative lane
        for i in range(n):
            k = "%s%d" % ("s" if side < 0 else "n", i)
            d_in, d_out = i * w, (n - 1 - i) * w
            if side < 0:    # near lane = positive lane, entered from P; far lane = negative lane, entered from Sn
                near, near_node, near_next = (x0, 0.0, 0.0), "P", "S"
                far, far_node, far_next = (x1, w, math.pi), "Sn", "Q"
            else:           # near lane = negative lane, entered from Sn; far lane = positive lane, entered from P
                near, near_node, near_next = (x1, w, math.pi), "Sn", "Q"
                far, far_node, far_next = (x0, 0.0, 0.0), "P", "S"
            # in from the near lane: [straight] -> right bend -> space
            e, src = near, near_node
            if d_in > 1e-9:       # (the in-block straights run along the main road's lanes: the same broken centre line on their left)
                e = net.add(src, k + "a", e, d_in, 0.0, 1, *NO, open_left=True)
                src = k + "a"
            e = net.add(src, k + "b", e, r * math.pi / 2, -1.0 / r, 1, *NO)
            space_start = e
            net.add(k + "b", k + "c", e, depth, 0.0, 1, LINE_CONTINUOUS, LINE_CONTINUOUS)
            dests.append(k + "c")
            # in from the far lane: [straight] -> left bend -> straight w -> space
            e, src = far, far_node
            if d_out > 1e-9:
                e = net.add(src, k + "d", e, d_out, 0.0, 1, *NO, open_left=True)
                src = k + "d"
            e = net.add(src, k + "e", e, r * math.pi / 2, 1.0 / r, 1, *NO)
            e = net.add(k + "e", k + "b", e, w, 0.0, 1, *NO)
            assert math.hypot(e[0] - space_start[0], e[1] - space_start[1]) < 1e-6 and abs(_wrap(e[2] - space_start[2])) < 1e-9
            # the space reversed = its spawn road (a parked vehicle faces the road)
            back = reverse(advance(space_start, depth, 0.0))
            e = net.add(k + "f", k + "g", back, depth, 0.0, 1, LINE_CONTINUOUS, LINE_CONTINUOUS)
            # out to the near lane: right bend -> [straight] -> the next block's road
            o = net.add(k + "g", k + "h" if d_out > 1e-9 else near_next, e, r * math.pi / 2, -1.0 / r, 1, *NO)
            if d_out > 1e-9:
                net.add(k + "h", near_next, o, d_out, 0.0, 1, *NO, open_left=True)
            # out to the far lane: straight w -> left bend -> [straight] -> the previous block's road
            o = net.add(k + "g", k + "i", e, w, 0.0, 1, *NO)
            o = net.add(k + "i", k + "j" if d_in > 1e-9 else far_next, o, r * math.pi / 2, 1.0 / r, 1, *NO)
            if d_in > 1e-9:
                net.add(k + "j", far_next, o, d_in, 0.0, 1, *NO, open_left=True)
    b = _Builder("parkinglot", net, 10, 40.0)
    half_slot = [RESPAWN_REGION_LONGITUDE / 2]
    for road in (("in0", "P"), ("inA", "Ai"), ("inB", "Bi")):         # entrants park in one of the spaces
        b.add_spawn_road(road, dests, half_slot, safe_only_first=False, exclusive=list(range(len(dests))) if unique_spaces else None)
    for side in ("s", "n"):                                          # parked vehicles leave through one of the three exits
        for i in range(n):
            b.add_spawn_road(("%s%df" % (side, i), "%s%dg" % (side, i)), ["end0", "endA", "endB"], half_slot, safe_only_first=False)
    return b.finish()


PG_BLOCK_WEIGHTS = (("C", 0.3), ("S", 0.3), ("X", 0.15), ("T", 0.15), ("O", 0.1))   # a draw per block of an int `sequence`


def _pgmap(sequence="SCS", seed=0, lanes=2, lane_width=LANE_WIDTH, lead=50.0):
    """Two-way road assembled from a block sequence, in the manner of MetaDrive's procedurally generated maps (the
    `MultiAgentMetaDrive` env of train_all_copo_dist.py:10,30; MetaDrive's `map` key = a block count or a block string):
      S  straight of 40-80 m (also stands in for the ramp blocks)
      C  curve of radius 30-60 m through 30-90 degrees to a random side
      X  standard intersection (radius 10): the road continues through a random arm -- right, straight or left
      T  T-intersection: as X with the straight arm missing
      O  roundabout (exit radius 10, inner radius 30, angle 70): the road continues through a random arm; the two driving
         directions take the two sides of the ring
    An int `sequence` draws that many blocks with MetaDrive's block weights.  Everything random comes from `seed`, so a
    (sequence, seed) pair names one map.  `lead` metres of straight road at both ends hold the spawn slots; vehicles enter at
    one end and leave at the other (the side arms of a junction block carry no traffic, so only the roads of the two routes
    are built).  Junction blocks are geometry along the routes: lane lines end inside them (the detectors see none)."""
    rng = np.random.RandomState(int(seed))
    if isinstance(sequence, (int, np.integer)):
        names, probs = zip(*PG_BLOCK_WEIGHTS)
        sequence = "".join(names[int(rng.choice(len(names), p=probs))] for _ in range(int(sequence)))
    w, n = lane_width, lanes
    net = Net(w)
    half = 10.0 + (2 * n - 1) * w / 2.0                     # intersection: stop line to junction centre (radius 10)
    ang = math.radians(70.0)                                # roundabout
    r_big = (2 * n - 1) * w + 30.0
    r_join = ((2 * n - 1) * w / 2.0 + 10.0) / math.cos(ang) - 10.0
    lane0 = lambda r_rightmost, left_turn: (r_rightmost - (n - 1) * w) if left_turn else (r_rightmost + (n - 1) * w)
    r_e0, r_b0, r_j0 = lane0(10.0, False), lane0(r_big, True), lane0(r_join, True)
    state = dict(pose=(0.0, 0.0, 0.0), heading=0.0, k=0)     # centre line (the yellow line) pose at the open end
    fwd_nodes, bwd_roads = ["f0"], []

    def centre_piece(ln, kap, lines=True):
        """One road per direction following a centre-line piece: forward lane 0 is w/2 to its right."""
        k, pose = state["k"], state["pose"]
        kw = {} if lines else dict(left_line=0, right_line=0, inner_line=0)
        a_off = -w / 2.0
        p0 = shift(pose, a_off)
        k0 = kap / (1.0 - kap * a_off) if kap else 0.0
        l0 = ln * (1.0 - kap * a_off) if kap else ln
        net.add("f%d" % k, "f%d" % (k + 1), p0, l0, k0, n, **kw)
        bwd_roads.append(("f%d" % k, "f%d" % (k + 1), "b%d" % (k + 1), "b%d" % k, kw))
        state["pose"], state["k"] = advance(pose, ln, kap), k + 1
        state["heading"] += kap * ln

    def ring_side(pose0, quarters):
        """Lane-0 pieces of one passage of the roundabout from an arm's entry (lane-0 pose `pose0` at the stop line)
        through `quarters` quarter turns to the exit arm's lane-0 pose."""
        pieces = [(r_e0 * ang, -1.0 / r_e0)]
        for q in range(quarters):
            pieces.append((r_b0 * (2 * ang - math.pi / 2), 1.0 / r_b0))
            if q < quarters - 1:
                pieces.append((r_j0 * (math.pi - 2 * ang), 1.0 / r_j0))
        pieces.append((r_e0 * ang, -1.0 / r_e0))
        return pieces

    centre_piece(lead, 0.0)
    for ch in str(sequence):
        if ch == "S":
            centre_piece(float(rng.uniform(40.0, 80.0)), 0.0)
        elif ch == "C":
            radius, a = float(rng.uniform(30.0, 60.0)), math.radians(float(rng.uniform(30.0, 90.0)))
            side = 1.0 if rng.rand() < 0.5 else -1.0
            if abs(state["heading"] + side * a) > math.radians(120.0):     # keep the chain from folding back over itself
                side = -side
            centre_piece(radius * a, side / radius)
        elif ch in "XT":
            arm = int(rng.choice([1, 2, 3] if ch == "X" else [1, 3]))       # 1 right, 2 straight, 3 left
            if arm != 2 and abs(state["heading"] + (math.pi / 2 if arm == 3 else -math.pi / 2)) > math.radians(120.0):
                arm = 4 - arm
            if arm == 2:
                centre_piece(2 * half, 0.0, lines=False)
            else:               # the lane arcs of a turn share the corner of the junction square as their centre: radius `half`
                centre_piece(half * math.pi / 2, (1.0 if arm == 3 else -1.0) / half, lines=False)
        elif ch == "O":
            arm = int(rng.choice([1, 2, 3]))
            turn = {1: -math.pi / 2, 2: 0.0, 3: math.pi / 2}[arm]
            if abs(state["heading"] + turn) > math.radians(120.0):
                arm, turn = 4 - arm, -turn
            k, pose = state["k"], state["pose"]
            # forward side: `arm` quarter turns; lane 0 of the forward road is w/2 right of the centre line
            p = shift(pose, -w / 2.0)
            fnodes = ["f%d" % k]
            for q, (ln, kap) in enumerate(ring_side(p, arm)):
                nxt = "f%d_%d" % (k, q)
                p_next = net.add(fnodes[-1], nxt, p, ln, kap, n, LINE_BROKEN if kap < 0 else LINE_CONTINUOUS, LINE_CONTINUOUS if kap < 0 else LINE_BROKEN)
                fnodes.append(nxt)
                p = p_next
            new_pose = shift(p, w / 2.0)                     # centre line at the exit arm
            # rename the last forward node to the chain's next node
            last = fnodes[-1]
            rd = net.roads.pop((fnodes[-2], last))
            net.roads[(fnodes[-2], "f%d" % (k + 1))] = rd
            net.solid[(fnodes[-2], "f%d" % (k + 1))] = net.solid.pop((fnodes[-2], last))
            net.adj[fnodes[-2]] = ["f%d" % (k + 1) if v == last else v for v in net.adj[fnodes[-2]]]
            net.adj.pop(last, None)
            net.adj.setdefault("f%d" % (k + 1), [])
            # backward side: from the exit arm back to the entry arm the other way round: 4 - arm quarter turns
            pb = reverse(shift(new_pose, w / 2.0))           # lane 0 of the backward carriageway at the exit arm, heading back
            bnodes = ["b%d" % (k + 1)]
            side = ring_side(pb, 4 - arm)
            for q, (ln, kap) in enumerate(side):
                nxt = ("b%d_%d" % (k, q)) if q < len(side) - 1 else "b%d" % k
                pb = net.add(bnodes[-1], nxt, pb, ln, kap, n, LINE_BROKEN if kap < 0 else LINE_CONTINUOUS, LINE_CONTINUOUS if kap < 0 else LINE_BROKEN)
                bnodes.append(nxt)
            want = reverse(shift(pose, w / 2.0))
            if math.hypot(pb[0] - want[0], pb[1] - want[1]) > 1e-6:
                raise AssertionError("roundabout block does not close: %r vs %r" % (pb, want))
            state["pose"], state["k"] = new_pose, k + 1
            state["heading"] += turn
        else:
            raise ValueError("pgmap block %r: S, C, X, T or O" % ch)
    centre_piece(lead, 0.0)
    for fa, fb, ba, bb, kw in reversed(bwd_roads):
        net.adverse(fa, fb, ba, bb, **kw)
    nb = state["k"]
    xe, ye = state["pose"][0], state["pose"][1]
    # centre the scene like the other maps: shift every road by minus the midpoint of the two ends
    ox, oy = -0.5 * xe, -0.5 * ye
    for key, (pp, ln, kap, nl) in list(net.roads.items()):
        net.roads[key] = ((pp[0] + ox, pp[1] + oy, pp[2]), ln, kap, nl)
    for ln_ in net.lines:
        ln_[0] += ox
        ln_[1] += oy
    b = _Builder("pgmap", net, 20, 0.5 * math.hypot(xe, ye) + lead)
    slots = spawn_slots(lead + ENTRANCE_LENGTH)
    b.add_spawn_road(("f0", "f1"), ["f%d" % nb], slots)
    b.add_spawn_road(("b%d" % nb, "b%d" % (nb - 1)), ["b0"], slots)
    return b.finish()


def pgmap(sequence="SCS", seed=0, **kw):
    """`_pgmap`; a DRAWN sequence (int) whose routes need more than MAX_SEGS roads -- several roundabouts -- is redrawn with
    its last roundabout replaced by a straight (still a function of (sequence, seed) alone)."""
    if not isinstance(sequence, (int, np.integer)):
        return _pgmap(sequence, seed, **kw)
    rng = np.random.RandomState(int(seed))
    names, probs = zip(*PG_BLOCK_WEIGHTS)
    seq = "".join(names[int(rng.choice(len(names), p=probs))] for _ in range(int(sequence)))
    while True:
        try:
            return _pgmap(seq, seed, **kw)
        except ValueError:
            if "O" not in seq:
                raise
            i = seq.rindex("O")
            seq = seq[:i] + "S" + seq[i + 1:]


MAP_BUILDERS = dict(intersection=intersection, roundabout=roundabout, tollgate=tollgate, parkinglot=parkinglot,
                    bottleneck=bottleneck, pgmap=pgmap)


def bounding_box(tables: MapTables, step=2.0):
    """(x_min, x_max, y_min, y_max) of the road network: lane-0 lines widened by their lanes.  Stands in for
    MetaDrive's `road_network.get_bounding_box()` (env_wrappers.py:268), used by the traffic-light columns."""
    lo, hi = np.array([np.inf, np.inf]), np.array([-np.inf, -np.inf])
    for r in range(tables.n_routes):
        pts = route_points(tables, r, step)
        pad = float(np.floor(tables.route_segs[r, :, SEG_LANES]).max()) * tables.lane_width
        lo, hi = np.minimum(lo, pts.min(0) - pad), np.maximum(hi, pts.max(0) + pad)
    return float(lo[0]), float(hi[0]), float(lo[1]), float(hi[1])


def ray_table(num_lasers, clockwise=True, offset_deg=0.0):
    """Unit vectors of detector beams in the vehicle frame (forward, left): beam 0 is turned `offset_deg` from the
    heading, beam k another k * 360 / n degrees -- CLOCKWISE by default: MetaDrive 0.2.5 adds k * step to the heading in
    its y-down frame (LiDAR: offset 0; side and lane-line detectors: offset 90)."""
    ang = math.radians(offset_deg) + 2.0 * np.pi * np.arange(num_lasers, dtype=np.float64) / num_lasers
    if clockwise:
        ang = -ang
    return np.stack([np.cos(ang), np.sin(ang)], axis=1).astype(np.float32)


def route_points(tables: MapTables, route: int, step=1.0, lateral=0.0):
    """Polyline of a route's lane-0 line (float64, for tests/plots); `lateral` metres to the left of it."""
    pts = []
    nseg = int(tables.route_meta[route, 1])
    for k in range(nseg):
        rec = tables.route_segs[route, k].astype(np.float64)
        pose = (rec[SEG_X0], rec[SEG_Y0], rec[SEG_TH0])
        for s in np.arange(0.0, rec[SEG_LEN], step):
            p = shift(advance(pose, s, rec[SEG_KAPPA]), lateral)
            pts.append((p[0], p[1]))
    rec = tables.route_segs[route, nseg].astype(np.float64)
    p = shift((rec[SEG_X0], rec[SEG_Y0], rec[SEG_TH0]), lateral)
    pts.append((p[0], p[1]))
    return np.asarray(pts)
