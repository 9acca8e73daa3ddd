"""Class re-exports, as roboticsPlayroomPybullet/envs/__init__.py does for the ids in scope."""
from .play_env import playEnv, pandaPick, pandaPush, UR5Reach, UR5PlayAbsRPY1Obj  # noqa: F401
from .play_env import UR5Play1Obj, UR5PlayRel1Obj, UR5PlayRelJoints1Obj, UR5PlayAbsJoints1Obj, UR5PlayRelRPY1Obj  # noqa: F401
from .play_env import pandaReach, pandaReach2D  # noqa: F401
from .play_env import (pandaPlay1Obj, pandaPlayRel1Obj, pandaPlayRelJoints1Obj, pandaPlayAbsJoints1Obj, pandaPlayAbsRPY1Obj,  # noqa: F401
                       pandaPlayRelRPY1Obj)
from .play_env import pandaPlay, pandaPlayRelJoints  # noqa: F401
