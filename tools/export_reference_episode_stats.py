#!/usr/bin/env python3
"""The reference's own record of PyBullet episodes: the last 100 training episodes (reward, length)
that stable-baselines3 kept in `ep_info_buffer` when backup_models/ppo_swing.zip was saved.

`data` inside the zip is JSON; the buffer in it is a base64 pickle of a deque of dicts
{'r': numpy float64 scalar, 'l': int, 't': float}. It is NOT unpickled: `pickletools.genops` only
disassembles the opcode stream (no opcode is executed, no object is built, nothing is imported); the
rewards are the 8 raw bytes of the SHORT_BINBYTES argument that follows each 'r' key, the lengths the
integer after each 'l' key. Output: tests/golden/ppo_swing_reference_episodes.json (a fixture: numbers
only). Runs where /root/reference exists; the fixture is what travels.
"""
import base64
import json
import os
import pickletools
import struct
import sys
import zipfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(ref="/root/reference"):
    z = zipfile.ZipFile(os.path.join(ref, "backup_models", "ppo_swing.zip"))
    data = json.loads(z.read("data"))
    raw = base64.b64decode(data["ep_info_buffer"][":serialized:"])
    rewards, lengths, times, key = [], [], [], None
    keys = {}  # memo index -> key string, for the BINGET references to 'r' / 'l' / 't'
    memo_next, last_str = 0, None
    for op, arg, _ in pickletools.genops(raw):
        name = op.name
        if name in ("SHORT_BINUNICODE", "BINUNICODE"):
            last_str = arg
            if arg in ("r", "l", "t"):
                key = arg
        elif name == "MEMOIZE":
            if last_str in ("r", "l", "t"):
                keys[memo_next] = last_str
            memo_next += 1
            last_str = None
            continue
        elif name in ("BINGET", "LONG_BINGET"):
            if arg in keys:
                key = keys[arg]
        elif name == "SHORT_BINBYTES" and len(arg) == 8 and key == "r":
            rewards.append(struct.unpack("<d", arg)[0])
            key = None
        elif name in ("BININT1", "BININT", "BININT2") and key == "l":
            lengths.append(int(arg))
            key = None
        elif name == "BINFLOAT" and key == "t":
            times.append(float(arg))
            key = None
        if name not in ("SHORT_BINUNICODE", "BINUNICODE"):
            last_str = None
    assert len(rewards) == len(lengths) == len(times) == 100, (len(rewards), len(lengths), len(times))
    out = {
        "source": "backup_models/ppo_swing.zip: data['ep_info_buffer'] (SB3 %s), disassembled with pickletools.genops, not unpickled" % z.read("_stable_baselines3_version").decode().strip(),
        "env": "SwingRacket-v0 under PyBullet, stochastic PPO policy during training (last 100 episodes before the save)",
        "num_timesteps": data["num_timesteps"], "n_steps": data["n_steps"], "episode_rewards": rewards, "episode_lengths": lengths,
        "episode_wallclock_s": times,
    }
    path = os.path.join(ROOT, "tests", "golden", "ppo_swing_reference_episodes.json")
    json.dump(out, open(path, "w"), indent=1)
    import statistics
    print("wrote %s: %d episodes, reward mean %.3f  stdev %.3f  min %.3f  max %.3f; lengths %s" % (
        path, len(rewards), statistics.mean(rewards), statistics.stdev(rewards), min(rewards), max(rewards), sorted(set(lengths))))


if __name__ == "__main__":
    main(*sys.argv[1:])
