"""tennisbot_rl_amd -- MI355X-native batched physics stepper for the SwingRacket-v0 /
Tennisbot-v0 environments of youliangtan/tennisbot-rl.

Importing the package does not touch the GPU or load the HIP library; that happens when
the first batch is created and fails loudly if the library or a device is missing.
"""
from .params import (ACT_DIM, ENV_SWING, ENV_TENNIS, F_AUTO_RESET, F_DEFAULT, F_NET, F_RACKET_BALL, F_RACKET_GROUND, OBS_DIM,  # noqa: F401
                     TbParams, default_params)

__version__ = "0.1.0"
