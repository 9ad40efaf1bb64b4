#!/usr/bin/env python3
"""Re-export the tensors of the reference's shipped PPO policy as a small .npz fixture.

backup_models/ppo_swing.zip -> policy.pth is read with torch.load(weights_only=True) (nothing
from the file is executed; the zip's `data` JSON with its embedded cloudpickle is not touched).
The output holds plain float32 arrays: a realistic, non-uniform action source for parity tests
and the warm start of train_swing.py --load-reference."""
import io
import os
import sys
import zipfile

import numpy as np
import torch

REF = os.environ.get("TB_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ppo_swing_policy.npz")


def main():
    z = zipfile.ZipFile(os.path.join(REF, "backup_models", "ppo_swing.zip"))
    sd = torch.load(io.BytesIO(z.read("policy.pth")), map_location="cpu", weights_only=True)
    arrays = {k.replace(".", "__"): v.numpy().astype(np.float32) for k, v in sd.items()}
    np.savez_compressed(OUT, **arrays)
    print("wrote", os.path.normpath(OUT), {k: v.shape for k, v in arrays.items()})


if __name__ == "__main__":
    sys.exit(main())
