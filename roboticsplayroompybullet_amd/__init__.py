"""MI355X-native batched playroom simulator behind the reference's gym surface (hot path only).

    import roboticsplayroompybullet_amd as rp
    env = rp.make('UR5PlayAbsRPY1Obj-v0')            # single env, reference surface (numpy in / out)
    vec = rp.VecPlayEnv('UR5PlayAbsRPY1Obj-v0', 4096) # batched, torch tensors on the GPU
"""
import importlib

from .vec_env import VecPlayEnv  # noqa: F401

# roboticsPlayroomPybullet/__init__.py:24,66,92 — the ids in scope, same entry-point style
_REGISTRY = {}


def register(id, entry_point, **kwargs):
    _REGISTRY[id] = (entry_point, kwargs)
    try:  # mirror into gym / gymnasium when they exist (they do not in the build image)
        from gym.envs.registration import register as gym_register  # type: ignore
        gym_register(id=id, entry_point=entry_point, **kwargs)
    except Exception:  # noqa: BLE001
        pass


def make(id, **kwargs):
    if id not in _REGISTRY:
        raise KeyError('%r is not registered (in scope: %s)' % (id, sorted(_REGISTRY)))
    entry_point, kw = _REGISTRY[id]
    mod, cls = entry_point.split(':')
    return getattr(importlib.import_module(mod), cls)(**dict(kw, **kwargs))


register(id='pandaPick-v0', entry_point='roboticsplayroompybullet_amd.envs:pandaPick')
register(id='pandaPush-v0', entry_point='roboticsplayroompybullet_amd.envs:pandaPush')     # __init__.py:19
register(id='UR5Reach-v0', entry_point='roboticsplayroompybullet_amd.envs:UR5Reach')
register(id='UR5PlayAbsRPY1Obj-v0', entry_point='roboticsplayroompybullet_amd.envs:UR5PlayAbsRPY1Obj')
# roboticsPlayroomPybullet/__init__.py:72,77,82,87,97 - the rest of the UR5 one-object play family
for _id, _cls in (('UR5Play1Obj-v0', 'UR5Play1Obj'), ('UR5PlayRel1Obj-v0', 'UR5PlayRel1Obj'), ('UR5PlayRelJoints1Obj-v0', 'UR5PlayRelJoints1Obj'),
                  ('UR5PlayAbsJoints1Obj-v0', 'UR5PlayAbsJoints1Obj'), ('UR5PlayRelRPY1Obj-v0', 'UR5PlayRelRPY1Obj')):
    register(id=_id, entry_point='roboticsplayroompybullet_amd.envs:' + _cls)
# roboticsPlayroomPybullet/__init__.py:4,9,33-64 - the Panda in default_scene and the Panda one-object play family
for _id, _cls in (('pandaReach-v0', 'pandaReach'), ('pandaReach2D-v0', 'pandaReach2D'), ('pandaPlay1Obj-v0', 'pandaPlay1Obj'),
                  ('pandaPlayRel1Obj-v0', 'pandaPlayRel1Obj'), ('pandaPlayRelJoints1Obj-v0', 'pandaPlayRelJoints1Obj'),
                  ('pandaPlayAbsJoints1Obj-v0', 'pandaPlayAbsJoints1Obj'), ('pandaPlayAbsRPY1Obj-v0', 'pandaPlayAbsRPY1Obj'),
                  ('pandaPlayRelRPY1Obj-v0', 'pandaPlayRelRPY1Obj')):
    register(id=_id, entry_point='roboticsplayroompybullet_amd.envs:' + _cls)
# roboticsPlayroomPybullet/__init__.py:29, 41 - the two-object play ids (the RP_WIDE build of the library)
register(id='pandaPlay-v0', entry_point='roboticsplayroompybullet_amd.envs:pandaPlay')
register(id='pandaPlayJoints-v0', entry_point='roboticsplayroompybullet_amd.envs:pandaPlayRelJoints')

__all__ = ['VecPlayEnv', 'make', 'register']
