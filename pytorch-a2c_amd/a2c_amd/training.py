"""Epoch driver with the reference's protocol (a2c/training.py:23-238) around the MI355X engine
(SURVEY.md section 8, row f1).  Same hyper-parameter keys as training_scripts/hyperparams.json; same
files in the save folder (``net.p``, ``best_net.p``, ``optim.p``, ``log.txt``); same loop:
wait for ``n_rollouts`` tokens on ``stop_q`` -> ``updater.update_model(shared_data)`` -> evaluation
rollout -> re-open the gate -> every 10 epochs save.

Differences forced by the design: ONE ``Runner`` (in a thread of this process) drives all
``n_envs`` envs in lock-step; the envs themselves are stepped by worker PROCESSES behind the pinned
pool region (``hostpool.ProcessEnvPool``, the counterpart of the reference's ``n_envs`` runner
processes, training.py:109-121), and runner and updater share the net object, so the weight publish
``net.load_state_dict(updater.net.state_dict())`` (training.py:165) disappears.  A failure inside
the runner thread or an env worker is re-raised here instead of dead-locking ``stop_q.get()``.  ``ml_utils`` (hyper-search CLI, un-vendored in the reference) is
not mirrored; the two helpers the loop needs (experiment numbering / save folder) are restated.
"""
import os
import queue
import threading
import time
from collections import deque

import numpy as np
import torch

from . import models, preprocessing
from .runner import HostEnvPool, Runner, SequentialEnvironment, StatsRunner
from .updater import Updater
from .utils import cuda_if, deque_maxmin, try_key

DEFAULTS = dict(n_frame_stack=4, n_rollouts=None, n_past_rews=25, h_size=256, lr=1e-4, lr_low=1e-12, lambda_=.98,
                gamma=.99, gamma_high=.995, val_coef=.5, entr_coef=.005, entr_coef_low=.001, pi_coef=1.0, max_norm=.5,
                resume=False, render=False, decay_lr=False, decay_entr=False, use_nstep_rets=False, norm_advs=True,
                use_bnorm=False, use_bptt=False, optim_type="RMSprop", n_test_eps=15, max_tsteps=4e7)


def get_exp_num(main_path, exp_name):
    """next free experiment number under main_path/exp_name (ml_utils.training.get_exp_num restated)"""
    folder = os.path.join(main_path, exp_name)
    nums = [int(d.rsplit("_", 1)[-1]) for d in (os.listdir(folder) if os.path.isdir(folder) else [])
            if d.rsplit("_", 1)[-1].isdigit()]
    return max(nums) + 1 if nums else 0


def get_save_folder(hyps):
    folder = os.path.join(hyps["main_path"], hyps["exp_name"], f"{hyps['exp_name']}_{hyps['exp_num']}")
    os.makedirs(folder, exist_ok=True)
    return folder


class _PositionalFactory:
    """env worker processes build env j as ``env_fn(j)`` whatever the callable names its parameter (the pool calls
    factories with keyword arguments)"""

    def __init__(self, fn):
        self.fn = fn

    def __call__(self, j):
        return self.fn(j)


def train(_, hyps, verbose=True, env_fn=None, eval_env=None, max_epochs=None, uniform_fn=None, on_epoch=None):
    """``hyps``: the reference's flat dict (training_scripts/hyperparams.json; ``n_rollouts`` need not be a multiple of
    ``n_envs``).  ``env_fn(j)`` (optional) builds env j as an object with ``reset()/step(a)`` returning already
    prepped frames -- otherwise gym envs are made from ``env_type`` / ``prep_fxn`` like the reference.  With the
    default process env pool ``env_fn`` travels to the worker processes pickled: lambdas and closures need
    ``cloudpickle`` (else pass a module-level callable, or ``hyps['env_pool'] = 'serial'``).  ``uniform_fn`` /
    ``on_epoch(epoch, updater, shared_data)`` are test hooks (sampler uniforms; called after every update).
    Returns the best evaluation reward."""
    hyps = dict(DEFAULTS, **hyps)
    if hyps["n_rollouts"] is None:
        hyps["n_rollouts"] = hyps["n_envs"]
    hyps["main_path"] = try_key(hyps, "main_path", "./")
    hyps["exp_num"] = get_exp_num(hyps["main_path"], hyps["exp_name"])
    save_folder = hyps["save_folder"] = get_save_folder(hyps)
    hyps["seed"] = try_key(hyps, "seed", int(time.time()))
    torch.manual_seed(hyps["seed"])
    np.random.seed(hyps["seed"])
    net_save_file, best_net_file = os.path.join(save_folder, "net.p"), os.path.join(save_folder, "best_net.p")
    optim_save_file, log_file = os.path.join(save_folder, "optim.p"), os.path.join(save_folder, "log.txt")
    log = open(log_file, "a" if hyps["resume"] else "w")
    for k in sorted(hyps):
        log.write(k + ":" + str(hyps[k]) + "\n")

    # environments
    serial = try_key(hyps, "env_pool", "process") == "serial"
    if env_fn is None:
        hyps["preprocessor"] = getattr(preprocessing, hyps["prep_fxn"])
        probe = eval_env or SequentialEnvironment(**hyps)             # the probe for shapes (training.py:60-65)
        hyps["is_discrete"], n_act = probe.is_discrete, probe.n
        kws = [dict({k: v for k, v in hyps.items() if k != "seed"}, seed=hyps["seed"] + j) for j in range(hyps["n_envs"])]
        if serial:
            envs = [SequentialEnvironment(**kw) for kw in kws]
            frame_shape = np.asarray(envs[0].prep_obs(np.zeros(envs[0].raw_shape, dtype=np.uint8))).shape
            pool = HostEnvPool(envs, frame_shape=frame_shape)
        else:
            from .hostpool import ProcessEnvPool
            # pong_prep yields {0,1} uint8 planes (preprocessing.py:15-16): one bit per pixel crosses the host link
            bits = try_key(hyps, "frame_bits", hyps["prep_fxn"] == "pong_prep")
            pool = ProcessEnvPool(SequentialEnvironment, hyps["n_envs"], env_kwargs=kws,
                                  n_workers=try_key(hyps, "n_env_workers", None), pong="Pong" in hyps["env_type"],
                                  action_shift=1 if hyps["env_type"] == "Pong-v0" else try_key(hyps, "action_shift", 0),
                                  frame_bits=bool(bits))
    else:
        if serial:
            pool = HostEnvPool([env_fn(j) for j in range(hyps["n_envs"])])
        else:
            from .hostpool import ProcessEnvPool
            pool = ProcessEnvPool(_PositionalFactory(env_fn), hyps["n_envs"], env_kwargs=[dict(j=j) for j in range(hyps["n_envs"])],
                                  n_workers=try_key(hyps, "n_env_workers", None), pong="Pong" in hyps["env_type"],
                                  action_shift=1 if hyps["env_type"] == "Pong-v0" else try_key(hyps, "action_shift", 0),
                                  probe_reset=True, frame_bits=bool(try_key(hyps, "frame_bits", False)))
        hyps["is_discrete"], n_act = True, hyps["action_size"]
    hyps["state_shape"] = [hyps["n_frame_stack"]] + list(pool.frame_shape[1:])
    if hyps["env_type"] == "Pong-v0":
        action_size, hyps["action_shift"] = 3, 1                      # training.py:67-69
    else:
        action_size, hyps["action_shift"] = n_act, try_key(hyps, "action_shift", 0)
    hyps["action_size"] = action_size
    shared_len = hyps["n_tsteps"] * hyps["n_rollouts"]
    if verbose:
        print("State Shape:,", hyps["state_shape"], "Num Samples Per Update:", shared_len)

    net = getattr(models, hyps["model"])(hyps["state_shape"], action_size, bnorm=hyps["use_bnorm"],
                                         **{k: v for k, v in hyps.items() if k not in ("use_bnorm",)})
    if hyps["resume"]:
        net.load_state_dict(torch.load(net_save_file))
    net = cuda_if(net)
    net._ensure_device()            # the flat parameter arena exists BEFORE the runner thread and the Updater touch it
    shared_data = {"states": cuda_if(torch.zeros((shared_len, *hyps["state_shape"]))),
                   "deltas": cuda_if(torch.zeros(shared_len)), "rewards": cuda_if(torch.zeros(shared_len)),
                   # (the reference keeps `actions` on the host, training.py:90,97; device resident here so that the
                   # rollout's device relay can write it -- nothing outside this function sees the dict)
                   "actions": cuda_if(torch.zeros(shared_len).long()), "dones": cuda_if(torch.zeros(shared_len))}
    if net.is_recurrent:
        shared_data["h_states"] = cuda_if(torch.zeros(shared_len, net.h_size))
    n_rollouts = hyps["n_rollouts"]
    gate_q, stop_q, reward_q = queue.Queue(n_rollouts), queue.Queue(n_rollouts), queue.Queue(1)
    reward_q.put(-1)
    runner = Runner(shared_data, hyps, gate_q, stop_q, reward_q, env_pool=pool, uniform_fn=uniform_fn)
    thread = threading.Thread(target=runner.run, args=(net,), daemon=True)
    thread.start()

    def collect():
        """stop_q.get() that notices a dead runner: a None token or a stopped thread re-raises its exception"""
        while True:
            try:
                tok = stop_q.get(timeout=1.0)
            except queue.Empty:
                if not thread.is_alive():
                    raise RuntimeError("the rollout thread stopped") from runner.error
                continue
            if tok is None:
                raise RuntimeError("the rollout thread failed") from runner.error
            return tok
    updater = Updater(net, hyps)
    if hyps["resume"]:
        updater.optim.load_state_dict(torch.load(optim_save_file))
    for i in range(n_rollouts):
        gate_q.put(i)
    # evaluation: the caller's single env (reference loop), else n_test_eps gym envs in lock-step on the device
    if eval_env is not None:
        stats_runner = StatsRunner(hyps, env=eval_env)
    else:
        stats_runner = StatsRunner(hyps) if env_fn is None else None
    entr_coef_diff, lr_diff = hyps["entr_coef"] - hyps["entr_coef_low"], hyps["lr"] - hyps["lr_low"]
    past_rews = deque([0] * hyps["n_past_rews"])
    best_eval_rew, epoch, T = -np.inf, 0, 0
    while T < hyps["max_tsteps"] and (max_epochs is None or epoch < max_epochs):
        basetime = time.time()
        epoch += 1
        for _i in range(n_rollouts):                                   # barrier: all slots collected
            collect()
        T += shared_len
        avg_reward = reward_q.get()
        reward_q.put(avg_reward)
        updater.update_model(shared_data)
        if on_epoch is not None:
            on_epoch(epoch, updater, shared_data)
        eval_rew = stats_runner.rollout(net) if stats_runner is not None else avg_reward
        if eval_rew > best_eval_rew:
            best_eval_rew = eval_rew
            updater.save_model(best_net_file, None)
        for i in range(n_rollouts):                                    # resume data collection
            gate_q.put(i)
        if hyps["decay_lr"]:
            updater.new_lr(max((1 - T / hyps["max_tsteps"]), 0) * lr_diff + hyps["lr_low"])
        if hyps["decay_entr"]:
            updater.entr_coef = entr_coef_diff * max((1 - T / hyps["max_tsteps"]), 0) + hyps["entr_coef_low"]
        if epoch % 10 == 0:
            updater.save_model(net_save_file, optim_save_file)
        past_rews.popleft()
        past_rews.append(avg_reward)
        max_rew, min_rew = deque_maxmin(past_rews)
        avg_action = shared_data["actions"].float().mean().item()
        if verbose:
            print("Epoch {} - T: {} -- {}".format(epoch, T, save_folder))
            updater.print_statistics()
        updater.log_statistics(log, T, avg_reward, avg_action, best_eval_rew)
        log.write("Grad Norm: {:.5f} – Avg Action: {:.5f} - Best EvalRew: {:.5f}\nPast Rews – High: {:.5f} - Low: {:.5f}\n"
                  "Time: {}\n\n".format(float(updater.norm), avg_action, best_eval_rew, max_rew, min_rew,
                                        time.time() - basetime))
    updater.save_model(net_save_file, optim_save_file)
    log.write("\nBestRew:" + str(best_eval_rew))
    log.close()
    for _i in range(n_rollouts):        # let the rollout the loop re-opened finish, then stop the thread and the env workers
        collect()
    gate_q.put(None)
    thread.join(timeout=30)
    runner.close()
    return best_eval_rew
