"""Class re-exports, as roboticsPlayroomPybullet/envs/__init__.py does for the ids in scope."""
from .play_env import playEnv, pandaPick, UR5Reach, UR5PlayAbsRPY1Obj  # noqa: F401
