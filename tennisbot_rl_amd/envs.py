"""Drop-in env surfaces over the batched stepper.

  * `SwingRacketEnv` / `TennisbotEnv`: the reference's gym-0.21 `Env` surface for ONE world
    (same constructor kwargs, spaces, return types and quirks) --
    tennisbot/envs/swingracket_env.py:24-192, tennisbot/envs/tennisbot_env.py:27-291.
  * `TennisbotVecEnv`: an SB3-`VecEnv`-shaped class over N worlds (numpy in / out, auto-reset,
    `terminal_observation` in infos) so `PPO("MlpPolicy", env, ...)` of train_swing.py:83-91
    / train.py:104-110 consumes it unchanged where stable-baselines3 is installed.
  * `make(id, **kwargs)` / `register_with_gym()`: the ids of tennisbot/__init__.py:3-11.

`gym` / `gymnasium` / `stable_baselines3` are optional (none is installed in the build
image): spaces fall back to a minimal `Box`, the VecEnv to a plain class with SB3's method
set. All arithmetic still happens in the HIP library; these classes only marshal.
"""
import warnings

import numpy as np

from .params import ENV_SWING, ENV_TENNIS
from .stepper import BatchedEnv

try:  # optional
    import gym as _gym  # noqa: F401
    from gym import spaces as _spaces
except Exception:  # pragma: no cover - depends on the host
    try:
        import gymnasium as _gym  # noqa: F401
        from gymnasium import spaces as _spaces
    except Exception:
        _gym, _spaces = None, None


class Box:
    """Minimal stand-in for gym.spaces.Box when gym is absent (same attributes)."""

    def __init__(self, low, high, dtype=np.float32):
        self.low = np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype)
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)
        self._rng = np.random.default_rng()

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)
        return [seed]

    def __repr__(self):
        return "Box(%s, %s, %s, %s)" % (self.low.min(), self.high.max(), self.shape, self.dtype)


def _box(low, high):
    low, high = np.array(low, dtype=np.float32), np.array(high, dtype=np.float32)
    if _spaces is not None:
        return _spaces.Box(low=low, high=high, dtype=np.float32)
    return Box(low, high)


# spaces, verbatim from the reference
SWING_ACTION = ([-1, -1.0, -1.0, -1, -1, -1], [1, 1.0, 1.0, 1, 1, 1])                      # swingracket_env.py:29-31
SWING_OBS = ([-20, -10, -20, -10, -15, -5], [20, 10, 20, 10, 0, 5])                        # swingracket_env.py:34-39
TENNIS_ACTION = ([-1.0, -1.0], [1.0, 1.0])                                                # tennisbot_env.py:43-44
TENNIS_OBS = ([-20, -20, -5, -5, -5, -5] + [-20, -20, 0, -10, -10, -10],
              [20, 20, 5, 5, 5, 5] + [20, 20, 10, 10, 10, 10])                             # tennisbot_env.py:51-55


def spaces_for(kind):
    if kind == ENV_SWING:
        return _box(*SWING_ACTION), _box(*SWING_OBS)
    return _box(*TENNIS_ACTION), _box(*TENNIS_OBS)


_Base = _gym.Env if _gym is not None else object


class _SingleEnv(_Base):
    """shared machinery of the two num_envs=1 facades; `done` is sticky until reset()"""
    metadata = {'render.modes': ['human']}
    _kind = None

    def __init__(self, use_gui=False, device=None, seed=0, params=None):
        if use_gui:
            warnings.warn("use_gui=True: the batched stepper has no GUI (PyBullet's GUI client is not reproduced)")
        self.action_space, self.observation_space = spaces_for(self._kind)
        self.np_random = np.random.default_rng()
        self._batch = BatchedEnv(self._kind, 1, device=device, seed=seed, params=params, auto_reset=False)
        self.done = False
        self.step_count = 0

    def _act(self, action):
        t = self._batch.torch
        a = np.asarray(action, dtype=np.float32).reshape(1, self._batch.act_dim)
        return t.from_numpy(a).to(self._batch.device)

    def seed(self, seed=None):
        """Reference behaviour (Appendix D.8): sets self.np_random, which nothing uses, and
        returns [seed]; reset draws are keyed by the constructor's `seed`."""
        self.np_random = np.random.default_rng(seed)
        return [seed]

    def close(self):
        self._batch.close()

    def render(self, mode='human'):
        pass

    def get_state(self):
        return self._batch.get_state()


class SwingRacketEnv(_SingleEnv):
    """SwingRacket-v0. tennisbot/envs/swingracket_env.py:24-192."""
    _kind = ENV_SWING

    def __init__(self, use_gui=False, delay_mode=False, device=None, seed=0, params=None):
        super().__init__(use_gui=use_gui, device=device, seed=seed, params=params)
        self.delay_mode = delay_mode  # accepted; the 1/240 s sleep is not reproduced
        self.reset()                  # the reference constructor resets (swingracket_env.py:61)

    @staticmethod
    def _obs(o):
        return tuple(float(x) for x in o)  # a 6-tuple of Python floats (Appendix D.1)

    def reset(self):
        o = self._batch.reset().cpu().numpy()[0]
        self.done, self.step_count = False, 0
        st = self._batch.get_state()
        self.goal = (float(st["goal"][0, 0]), float(st["goal"][0, 1]))
        self.spawn_pos = [float(x) for x in st["spawn_pos"][0]]
        self.initial_dist_to_goal = float(st["init_dist"][0])
        return self._obs(o)

    def step(self, action):
        obs, rew, done = self._batch.step(self._act(action))
        self.done = bool(done.item())
        self.step_count += int(self._batch.last_substeps().item())
        return self._obs(obs.cpu().numpy()[0]), float(rew.item()), self.done, dict()


class TennisbotEnv(_SingleEnv):
    """Tennisbot-v0. tennisbot/envs/tennisbot_env.py:27-291."""
    _kind = ENV_TENNIS

    def __init__(self, use_gui=False, is_sparse_reward=False, device=None, seed=0, params=None):
        super().__init__(use_gui=use_gui, device=device, seed=seed, params=params)
        self.is_sparse_reward = is_sparse_reward  # stored, never read (Appendix D.7)
        self.racket_scale = 1.0
        self.reset()  # tennisbot_env.py:88

    def set_racket_scale(self, scale):
        """tennisbot_env.py:213-215: used by the next reset()"""
        self.racket_scale = scale

    def reset(self):
        if self._batch.params.racket_scale != np.float32(self.racket_scale):
            self._batch.set_racket_scale(self.racket_scale)
        o = self._batch.reset().cpu().numpy()[0]
        self.done, self.step_count = False, 0
        return o.astype(np.float32)

    def step(self, action):
        obs, rew, done = self._batch.step(self._act(action))
        self.done = bool(done.item())
        self.step_count += 1
        return obs.cpu().numpy()[0].astype(np.float32), float(rew.item()), self.done, dict()


_REGISTRY = {"SwingRacket-v0": SwingRacketEnv, "Tennisbot-v0": TennisbotEnv}


def make(env_id, **kwargs):
    """gym.make counterpart for the two ids of tennisbot/__init__.py:3-11"""
    return _REGISTRY[env_id](**kwargs)


def register_with_gym():
    """register the ids with gym / gymnasium when one of them is installed"""
    if _gym is None:
        return False
    from gym.envs.registration import register  # type: ignore
    for env_id, cls in _REGISTRY.items():
        try:
            register(id=env_id, entry_point="tennisbot_rl_amd.envs:%s" % cls.__name__)
        except Exception:
            pass
    return True


try:  # optional: real SB3 base class when available so isinstance checks pass
    from stable_baselines3.common.vec_env import VecEnv as _VecEnvBase  # type: ignore
except Exception:  # pragma: no cover - depends on the host
    _VecEnvBase = object


class TennisbotVecEnv(_VecEnvBase):
    """SB3 VecEnv surface (SB3 1.8.0, the version of backup_models/ppo_swing.zip) over N
    worlds on one GPU: numpy in / out, auto-reset, infos[i]['terminal_observation'].
    `tensor_step` exposes the zero-copy device path underneath."""

    def __init__(self, env_id, num_envs, device=None, seed=0, env_id_base=0, params=None):
        kind = {"SwingRacket-v0": ENV_SWING, "Tennisbot-v0": ENV_TENNIS}[env_id]
        self.env_id = env_id
        self.batch = BatchedEnv(kind, num_envs, device=device, seed=seed, env_id_base=env_id_base, params=params, auto_reset=True)
        action_space, observation_space = spaces_for(kind)
        if _VecEnvBase is object:
            self.num_envs, self.observation_space, self.action_space = int(num_envs), observation_space, action_space
        else:
            super().__init__(int(num_envs), observation_space, action_space)
        self.metadata = {'render.modes': ['human']}
        self._actions = None
        self.racket_scale = 1.0

    # ---- VecEnv API
    def reset(self):
        return self.batch.reset().cpu().numpy()

    def step_async(self, actions):
        t = self.batch.torch
        a = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.num_envs, self.batch.act_dim)
        self._actions = t.from_numpy(a).to(self.batch.device)

    def step_wait(self):
        obs, rew, done = self.batch.step(self._actions)
        obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy().astype(bool)
        infos = [{} for _ in range(self.num_envs)]
        if done.any():
            term = self.batch.terminal_obs().cpu().numpy()
            for i in np.nonzero(done)[0]:
                infos[i]["terminal_observation"] = term[i]
        return obs, rew, done, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def tensor_step(self, actions):
        """device tensors in, device tensors out, no host synchronisation"""
        return self.batch.step(actions)

    def close(self):
        self.batch.close()

    def seed(self, seed=None):
        return [seed] * self.num_envs

    def render(self, mode='human'):
        pass

    def get_images(self):
        return []

    def _indices(self, indices):
        if indices is None:
            return list(range(self.num_envs))
        return [indices] if isinstance(indices, int) else list(indices)

    def get_attr(self, attr_name, indices=None):
        return [getattr(self, attr_name) for _ in self._indices(indices)]

    def set_attr(self, attr_name, value, indices=None):
        setattr(self, attr_name, value)

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        out = getattr(self, method_name)(*method_args, **method_kwargs)
        return [out for _ in self._indices(indices)]

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False for _ in self._indices(indices)]

    def set_racket_scale(self, scale):
        """curriculum hook of train.py:147-149,164-176 (`env.set_racket_scale(s)`): applies
        to each env at its own next reset, as in tennisbot_env.py:230-234"""
        self.racket_scale = scale
        self.batch.set_racket_scale(scale)
