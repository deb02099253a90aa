"""Stand-in for `mujoco_py.MjSim` / `load_model_from_path`, used ONLY by gen_maze_ref_golden.py so that the
REFERENCE's env/maze.py can be imported and run in this container (MuJoCo 1.50 and mujoco_py are absent).

It is not MuJoCo: it implements the documented kinematic surrogate of DESIGN.md section 6 behind the few members
env/maze.py touches -- `sim.data.qpos / qvel / ctrl / ncon`, `sim.model.geom_pos`, `sim.step()`, `sim.forward()` -- so
that everything AROUND the physics (step / reset / expert / distance / offline-data control flow, rewards,
termination, reset ranges, the wall moves of reset()) is executed by the reference's own code.

Surrogate (one env step = the 500 `sim.step()` calls of env/maze.py:146-147):
  * the model is read from the reference's own env/assets/simple_maze.xml: 11 geoms in document order (5..8 are the
    walls reset() moves), box half-extents mapped through `zaxis`, tool radius, joint range, motor gear, joint damping;
  * a body at rest (qvel == 0, which env/maze.py:141,150 guarantees at the start of every burst) that is not in contact
    moves along the straight segment GAIN * ctrl, scanned in 64 equal sub-steps, and stops at the first sub-step
    where the disc touches a wall rectangle or an arena plane; the remaining `step()` calls of the burst find
    qvel != 0 and only refresh the contact count;
  * GAIN is the displacement per unit control of 500 semi-implicit Euler steps (dt = 0.002) of a 0.09817 kg body with
    motor gear 0.05 and joint damping 0.01, started at rest;
  * ncon = 3 (the cylinder resting on the ground plane) + 1 when the disc touches a wall or an arena plane.
All arithmetic is Python float (IEEE double, no FMA) in the same operation order as oracle/rrl_oracle.c.
"""
import math
import xml.etree.ElementTree as ET

import numpy as np

GAIN = 0.24667750873451577
SUBSTEPS = 64
DT, N_SIM_STEPS, DENSITY = 0.002, 500, 1000.0


def _vec(text, n=3):
    v = [float(t) for t in text.split()]
    return v + [0.0] * (n - len(v))


class StubModel:
    def __init__(self, path):
        root = ET.parse(path).getroot()
        world = root.find("worldbody")
        geoms = []
        for el in world.iter("geom"):                       # document order = MuJoCo geom ids
            geoms.append(el)
        self.geom_names = [g.get("name") for g in geoms]
        self.geom_pos = np.array([_vec(g.get("pos", "0 0 0")) for g in geoms])
        self.geom_type = [g.get("type", "sphere") for g in geoms]
        self.geom_size = [_vec(g.get("size", "0")) for g in geoms]
        self.geom_zaxis = [_vec(g.get("zaxis", "0 0 1")) for g in geoms]
        tool = world.find("body")
        tg = tool.find("geom")
        self.radius = _vec(tg.get("size"))[0]
        half_height = _vec(tg.get("size"))[1]
        joints = tool.findall("joint")
        self.joint_range = _vec(joints[0].get("range"), 2)
        self.damping = float(joints[0].get("damping"))
        self.gear = float(root.find("default").find("motor").get("gear"))
        self.mass = DENSITY * math.pi * self.radius ** 2 * (2 * half_height)
        self.wall_ids = [i for i, t in enumerate(self.geom_type) if t == "box"]
        self.plane_ids = [i for i, (t, n) in enumerate(zip(self.geom_type, self.geom_names))
                          if t == "plane" and n != "ground"]

    def free_gain(self):
        """Displacement per unit control of the 500-step burst from rest (semi-implicit Euler)."""
        v = x = 0.0
        for _ in range(N_SIM_STEPS):
            v = v + DT * (self.gear - self.damping * v) / (self.mass + DT * self.damping)
            x = x + DT * v
        return x

    def wall_half_extents(self, gid):
        """World-frame half extents (x, y) of a box whose local z axis points along world -x (zaxis='-1 0 0'):
        local z -> world x, local y -> world y."""
        sx, sy, sz = self.geom_size[gid]
        assert self.geom_zaxis[gid] == [-1.0, 0.0, 0.0]
        return sz, sy


def load_model_from_path(path):
    return StubModel(path)


class _Data:
    def __init__(self):
        self.qpos = np.zeros(2)
        self.qvel = np.zeros(2)
        self.ctrl = np.zeros(2)
        self.ncon = 3


class MjSim:
    def __init__(self, model):
        self.model = model
        self.data = _Data()
        assert abs(model.free_gain() - GAIN) < 1e-12 * GAIN, (model.free_gain(), GAIN)
        self.forward()

    # -- geometry ----------------------------------------------------------------------------------
    def _contact(self, x, y):
        m = self.model
        lo, hi = m.joint_range
        r = m.radius
        # arena planes at the joint limits (simple_maze.xml:16-19): +x wall at hi, -x wall at lo, same for y
        if hi - x <= r or x - lo <= r:
            return True
        if hi - y <= r or y - lo <= r:
            return True
        for gid in m.wall_ids:
            cx, cy = m.geom_pos[gid][0], m.geom_pos[gid][1]
            hx, hy = m.wall_half_extents(gid)
            dx = abs(x - cx) - hx
            dy = abs(y - cy) - hy
            dx = 0.0 if dx < 0.0 else dx
            dy = 0.0 if dy < 0.0 else dy
            if dx * dx + dy * dy <= r * r:
                return True
        return False

    def forward(self):
        x, y = float(self.data.qpos[0]), float(self.data.qpos[1])
        self.data.ncon = 3 + int(self._contact(x, y))

    def step(self):
        d = self.data
        if not d.qvel.any():                                # at rest: the whole burst's displacement
            x0, y0 = float(d.qpos[0]), float(d.qpos[1])
            lo, hi = self.model.joint_range
            if not self._contact(x0, y0):
                dx, dy = GAIN * float(d.ctrl[0]), GAIN * float(d.ctrl[1])
                qx, qy = x0, y0
                for k in range(1, SUBSTEPS + 1):
                    f = float(k) * (1.0 / SUBSTEPS)
                    qx = min(max(x0 + dx * f, lo), hi)
                    qy = min(max(y0 + dy * f, lo), hi)
                    if self._contact(qx, qy):
                        break
                d.qpos[0], d.qpos[1] = qx, qy
            d.qvel[:] = 1.0                                 # moving: the rest of the burst changes nothing
        self.forward()
