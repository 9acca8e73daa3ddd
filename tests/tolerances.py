"""Tolerances shared by the GPU parity tests.

THE GRIPPER'S LIMIT CHATTER.  Since round 3 the shipped model has Bullet's joint-limit rule (btMultiBodyJointLimitConstraint::createConstraintRows: a
row only while the limit is violated, erp 0.2).  A gripper joint whose position motor is commanded past its limit - every "open" action does that to
the Robotiq's six joints and to the Panda's two fingers (environments.py:1037-1073) - then runs a sawtooth: inside the violation the limit row wins
and the joint creeps back towards the limit by a factor 0.8 per substep; when rounding noise carries it across, the row is gone, the motor kicks
the joint back in by (max impulse / inertia) x dt, and so on.  Amplitude: 1.1 mm for a Robotiq pad (100 N x dt on a 1 kg link: the observation
obs_quat['gripper'] = 23 x that = 0.026), up to 0.1 x (target - limit) for the light Robotiq fingers (0.1 rad) and the Panda's fingers (4 mm).
The PHASE of the sawtooth is decided at rounding level, so two fp32 evaluation orders (device, fp32 CPU oracle) or fp32 vs fp64 disagree by up to
the amplitude on those joints and on the gripper observation while agreeing to 1e-6 on everything else.  The arm's own joints are held to the
tests' tight tolerances; the gripper's joints and the gripper observation to the amplitudes below."""

GRIP_OBS_TOL = {'ur5': 0.03, 'panda': 0.005}          # obs_quat's gripper entry: UR5 q18 * 23, Panda q9 [m]
GRIP_OBS_REST_TOL = {'ur5': 2e-3, 'panda': 2e-4}      # the same with no position motor pushing (after reset)
GRIP_JOINT_TOL = 0.15                                 # gripper joints, rad or m, max(1, |q|)-relative like the joint measure: the measured sawtooth (0.094 - 0.096 over the 200-step rollouts of round 4) + 50 %
N_MAIN = {'U': 6, 'R': 6, 'P': 7, 'Q': 7, 'V': 7, 'W': 7}     # the arm's own joints (chain to the EE link); the dofs after them are the gripper's
ARM = {'U': 'ur5', 'R': 'ur5', 'P': 'panda', 'Q': 'panda', 'V': 'panda', 'W': 'panda'}
GRIP_INDEX = {'U': 7, 'R': 6, 'P': 6, 'Q': 6, 'V': 7, 'W': 7}  # position of the gripper entry in obs_quat


def obs_atol(kind, n, base, rest=False):
    """per-component absolute tolerance for an obs_quat-like vector of length n"""
    import numpy as np
    t = np.full(n, float(base))
    g = GRIP_INDEX[kind]
    if g < n:
        t[g] = max(base, (GRIP_OBS_REST_TOL if rest else GRIP_OBS_TOL)[ARM[kind]])
    return t


class Followers:
    """The CPU oracles that follow one env of a device test: the fp64 one, the fp32 one, and `extra` more fp32 ones whose arm joints start
    NUDGE (1e-5, relative) off.  Where a comparison crosses events that rounding decides (which substep a gripper joint crosses its limit in,
    a contact that makes or breaks, the iteration at which the IK's residual test stops it), the device - one more evaluation order of the
    same fp32 arithmetic - is held to three times the largest distance of any fp32 follower from the fp64 one; everywhere else to the test's
    plain tolerance.

    Why 1e-5 and not an ulp: the rows of the light Robotiq links are ill-conditioned (a position motor and a violated limit fighting over a
    link of 1e-4 kg m^2 on a 20 kg arm), and two fp32 evaluation orders of ONE substep from the SAME state differ by up to 1e-3 relative in
    those links' velocities (tools/gpu_bisect3.py: device vs fp32 CPU oracle 4e-2 rad/s, device vs fp64 CPU oracle 4e-3 rad/s on 10 rad/s),
    which reaches the arm's own joints at the 1e-5 level within a step.  An fp32 CPU run shares the fp64 oracle's evaluation order and
    underestimates that; the nudged runs stand in for it.

    How many: an event's outcome is one draw per run.  The 200-step tests follow every env with `extra` = 6 nudged runs (+-1e-5, +-2e-5, +-3e-5): with two,
    2 of 64 envs had an event (the gripper striking the block at 4 mm per substep; an IK stopping one iteration apart) that the
    device drew and none of the three CPU runs did."""
    NUDGE = 1e-5


    def __init__(self, kind, seed, env_index, extra=2, **kw):
        from oracle import OracleEnv
        self.o64 = OracleEnv(kind, seed=seed, env_index=env_index, **kw)
        self.o32 = OracleEnv(kind, seed=seed, env_index=env_index, f32=True, **kw)
        self.more = [OracleEnv(kind, seed=seed, env_index=env_index, f32=True, **kw) for _ in range(extra)]

    def all(self):
        return [self.o64, self.o32] + self.more

    def reset(self):
        """every follower resets on its own (the fp64 one and the fp32 ones then differ by what the reset's 100 settle substeps make of rounding);
        the extra ones get their arm joints nudged afterwards.  Returns (fp32 obs, fp64 obs)."""
        a64 = self.o64.reset()
        a32 = self.o32.reset()
        for o in self.more:
            o.reset()
        self.nudge()
        return a32, a64

    def start_from(self, src):
        """every follower takes the state (and goal) of `src`, an oracle env that has been reset"""
        import ctypes as C
        import numpy as np
        s = src.get_state()
        g = np.ascontiguousarray(src.calc_state()['desired_goal'], dtype=np.float64)
        src.clear_quat_memory()
        for o in self.all():
            if o is not src:
                o.reset()
                o.set_state(s)
                o.lib.rpo_set_goal(o.h, g.ctypes.data_as(C.POINTER(C.c_double)))
        self.nudge()

    def start_from_state(self, s):
        """every follower resets on its own and then takes the state vector `s` (OracleEnv.get_state layout); returns (fp32 obs, fp64 obs) of the resets"""
        res = []
        for o in self.all():
            res.append(o.reset())
            o.set_state(s)
        self.nudge()
        return res[1], res[0]

    def nudge(self):
        for k, o in enumerate(self.more):
            s = o.get_state()
            s[:o.n_arm] *= 1.0 + (self.NUDGE if k % 2 == 0 else -self.NUDGE) * (1 + k // 2)
            o.set_state(s)

    def step(self, a):
        """steps all followers; returns (fp32 result, fp64 result, list of all results)"""
        res = [o.step(a) for o in self.all()]
        return res[1], res[0], res

    def gap(self, get):
        """largest distance of an fp32 follower from the fp64 one in get(oracle) (an array)"""
        import numpy as np
        ref = np.asarray(get(self.o64), dtype=np.float64)
        return np.max([np.abs(np.asarray(get(o), dtype=np.float64) - ref) for o in [self.o32] + self.more], axis=0)
