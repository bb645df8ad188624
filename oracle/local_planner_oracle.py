"""TEST INFRASTRUCTURE ONLY (see oracle/README or DESIGN.md section 3): CPU restatement of the control cycle around
DWAPlanner::findBestPath -- DWAPlannerROS::setPlan / computeVelocityCommands / isGoalReached
(dwa_local_planner/src/dwa_planner_ros.cpp:130-158,176-300), LocalPlannerUtil::getLocalPlan
(base_local_planner/src/local_planner_util.cpp:105-123), goal_functions.cpp (prunePlan :69-86,
transformGlobalPlan :88-174, getGoalPose :175-214, stopped :248-253) and LatchedStopRotateController
(src/latched_stop_rotate_controller.cpp:37-273).

Pure Python on top of the C++ DWA oracle (pyoracle.DwaPlanner).  Poses are (x, y, yaw); the tf lookup is an
optional planar transform.  tf's 3-D arithmetic and the `angles` package are NOT under /root/reference:
angles::normalize_angle is restated in its fmod form -- parity unpinned for that part (no reference test
covers these functions)."""
import math

import numpy as np

BRANCH_NONE, BRANCH_DWA, BRANCH_STOP, BRANCH_ROTATE, BRANCH_AT_GOAL = 0, 1, 2, 3, 4


def normalize_angle_positive(a):  # angles.h
    return math.fmod(math.fmod(a, 2.0 * math.pi) + 2.0 * math.pi, 2.0 * math.pi)


def normalize_angle(a):
    r = normalize_angle_positive(a)
    if r > math.pi:
        r -= 2.0 * math.pi
    return r


def shortest_angular_distance(frm, to):
    return normalize_angle(to - frm)


def sign(x):  # base_local_planner/include/base_local_planner/goal_functions.h
    return -1.0 if x < 0.0 else 1.0


def transform_global_plan(plan, pose, T, dist_threshold):
    """goal_functions.cpp:88-174 on an (n, 3) plan; returns the local plan (m, 3) or None for an empty plan."""
    if len(plan) == 0:
        return None
    rx, ry = pose[0], pose[1]
    if T is not None:
        c, s = math.cos(T[2]), math.sin(T[2])
        dx, dy = pose[0] - T[0], pose[1] - T[1]
        rx, ry = c * dx + s * dy, -s * dx + c * dy
    thr = dist_threshold * dist_threshold
    i, sq = 0, 0.0
    while i < len(plan):  # :126-134
        xd, yd = rx - plan[i][0], ry - plan[i][1]
        sq = xd * xd + yd * yd
        if sq <= thr:
            break
        i += 1
    out = []
    while i < len(plan) and sq <= thr:  # :140-154
        px, py, pth = plan[i]
        if T is not None:
            out.append((c * px - s * py + T[0], s * px + c * py + T[1], pth + T[2]))
        else:
            out.append((px, py, pth))
        xd, yd = rx - px, ry - py
        sq = xd * xd + yd * yd
        i += 1
    return out


def prune_plan(pose, plan, global_plan):
    """goal_functions.cpp:69-86: erases the leading poses of BOTH lists in lockstep."""
    while plan:
        xd, yd = pose[0] - plan[0][0], pose[1] - plan[0][1]
        if xd * xd + yd * yd < 1:
            break
        plan.pop(0)
        global_plan.pop(0)


def stopped(odom_vel, rot_stopped, trans_stopped):  # goal_functions.cpp:248-253
    return abs(odom_vel[2]) <= rot_stopped and abs(odom_vel[0]) <= trans_stopped and abs(odom_vel[1]) <= trans_stopped


class DwaPlannerRos:
    """One robot's DWAPlannerROS.  `planner` is a pyoracle.DwaPlanner bound to the robot's costmap."""

    def __init__(self, planner, limits, size_x, size_y, resolution, footprint):
        self.dp, self.lim, self.fp = planner, dict(limits), np.asarray(footprint, np.float64)
        self.dist_threshold = max(size_x * resolution / 2.0, size_y * resolution / 2.0)  # goal_functions.cpp:119-120
        self.global_plan, self.T = [], None
        self.xy_tolerance_latch = False
        self.rotating_to_goal = False

    def set_plan(self, plan_xyyaw, T=None):  # dwa_planner_ros.cpp:130-140
        self.xy_tolerance_latch = False
        self.global_plan = [tuple(float(v) for v in p) for p in plan_xyyaw]
        self.T = None if T is None else tuple(float(v) for v in T)
        self.dp.set_plan()
        return True

    def goal(self):  # getGoalPose
        if not self.global_plan:
            return None
        g = self.global_plan[-1]
        if self.T is None:
            return g
        c, s = math.cos(self.T[2]), math.sin(self.T[2])
        return (c * g[0] - s * g[1] + self.T[0], s * g[0] + c * g[1] + self.T[1], g[2] + self.T[2])

    def is_goal_reached(self, pose, odom_vel):  # latched_stop_rotate_controller.cpp:66-109
        lim = self.lim
        g = self.goal()
        if g is None:
            return False
        if (lim["latch_xy_goal_tolerance"] and self.xy_tolerance_latch) or \
                math.hypot(g[0] - pose[0], g[1] - pose[1]) <= lim["xy_goal_tolerance"]:
            if lim["latch_xy_goal_tolerance"] and not self.xy_tolerance_latch:
                self.xy_tolerance_latch = True
            angle = shortest_angular_distance(pose[2], g[2])
            if abs(angle) <= lim["yaw_goal_tolerance"] and stopped(odom_vel, lim["rot_stopped_vel"], lim["trans_stopped_vel"]):
                return True
        return False

    def compute_velocity_commands(self, pose, odom_vel, have_pose=True):
        """dwa_planner_ros.cpp:252-300 -> dict(ok, cmd, branch, local_plan_points, trajectory_points)."""
        lim = self.lim
        r = dict(ok=False, cmd=(0.0, 0.0, 0.0), branch=BRANCH_NONE, local_plan_points=0, trajectory_points=0)
        if not have_pose:
            return r
        local = transform_global_plan(self.global_plan, pose, self.T, self.dist_threshold)
        if local is None:
            return r
        if lim["prune_plan"]:
            prune_plan(pose, local, self.global_plan)
        r["local_plan_points"] = len(local)
        if not local:
            return r
        pos32 = np.asarray(pose, np.float32)
        vel32 = np.asarray(odom_vel, np.float32)
        plan_xy = np.asarray([(p[0], p[1]) for p in local], np.float64)
        self.dp.update_plan(pos32, plan_xy)  # always, :274
        g = self.goal()
        reached = False  # isPositionReached :37-58
        if g is not None and ((lim["latch_xy_goal_tolerance"] and self.xy_tolerance_latch) or
                              math.hypot(g[0] - pose[0], g[1] - pose[1]) <= lim["xy_goal_tolerance"]):
            self.xy_tolerance_latch = True
            reached = True
        if not reached:  # dwaComputeVelocityCommands :176-247
            res, traj, _, _, _ = self.dp.cycle(pos32, vel32, plan_xy, self.fp, want_samples=False)
            r["branch"] = BRANCH_DWA
            r["cmd"] = (res.drive[0], res.drive[1], res.drive[2])
            r["ok"] = res.cost >= 0
            r["trajectory_points"] = res.n_points if r["ok"] else 0
            return r
        # computeVelocityCommandsStopRotate :211-273
        if g is None:
            return r
        if lim["latch_xy_goal_tolerance"] and not self.xy_tolerance_latch:
            self.xy_tolerance_latch = True
        yaw, vel_yaw = pose[2], odom_vel[2]
        angle = shortest_angular_distance(yaw, g[2])
        if abs(angle) <= lim["yaw_goal_tolerance"]:
            self.rotating_to_goal = False
            r.update(ok=True, branch=BRANCH_AT_GOAL)
            return r
        acc = (lim["acc_lim_x"], lim["acc_lim_y"], lim["acc_lim_theta"])
        sp = lim["sim_period"]
        if not self.rotating_to_goal and not stopped(odom_vel, lim["rot_stopped_vel"], lim["trans_stopped_vel"]):
            vx = sign(odom_vel[0]) * max(0.0, abs(odom_vel[0]) - acc[0] * sp)  # stopWithAccLimits :111-146
            vy = sign(odom_vel[1]) * max(0.0, abs(odom_vel[1]) - acc[1] * sp)
            vth = sign(vel_yaw) * max(0.0, abs(vel_yaw) - acc[2] * sp)
            r["branch"] = BRANCH_STOP
            if self.dp.check_trajectory(pos32, vel32, np.asarray((vx, vy, vth), np.float32)):
                r.update(ok=True, cmd=(vx, vy, vth))
            return r
        self.rotating_to_goal = True  # rotateToGoal :148-209
        ang_diff = angle
        v = min(lim["max_rot_vel"], max(lim["min_rot_vel"], abs(ang_diff)))
        max_acc_vel = abs(vel_yaw) + acc[2] * sp
        min_acc_vel = abs(vel_yaw) - acc[2] * sp
        v = min(max(abs(v), min_acc_vel), max_acc_vel)
        v = min(math.sqrt(2 * acc[2] * abs(ang_diff)), abs(v))
        v = min(lim["max_rot_vel"], max(lim["min_rot_vel"], v))
        if ang_diff < 0:
            v = -v
        r["branch"] = BRANCH_ROTATE
        if self.dp.check_trajectory(pos32, vel32, np.asarray((0.0, 0.0, v), np.float32)):
            r.update(ok=True, cmd=(0.0, 0.0, v))
        return r
