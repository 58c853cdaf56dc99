"""Drop-in for the reference's ``a2c/utils.py`` (same names, same argument meaning), with the
arithmetic on the MI355X kernels.

=================  ====================================  =================================
function           reference                             runs on
=================  ====================================  =================================
``discount``       utils.py:63-79                        a2c_discount_scan (HIP, bit-exact)
``sample_action``  utils.py:45-60                        a2c_sample_probs (HIP)
``next_state``     utils.py:26-43                        host (numpy deque bookkeeping; the
                                                         batched device form is
                                                         ops.frame_stack_push)
``cuda_if``        utils.py:9-12                         --
``try_key``        utils.py:4-7                          --
``deque_maxmin``   utils.py:14-24                        --
=================  ====================================  =================================
"""
import numpy as np
import torch

from . import ops


def try_key(dict_, key, default):
    return dict_[key] if key in dict_ else default


def cuda_if(tobj):
    if torch.cuda.is_available():
        tobj = tobj.cuda()
    return tobj


def deque_maxmin(deq):
    hi = lo = deq[0]
    for v in deq:
        if v > hi:
            hi = v
        if v < lo:
            lo = v
    return hi, lo


def _dev(t):
    if not torch.cuda.is_available():
        raise RuntimeError("a2c_amd needs a HIP device: there is no CPU fallback for its kernels")
    if isinstance(t, np.ndarray):
        t = torch.from_numpy(t)
    elif not torch.is_tensor(t):
        t = torch.as_tensor(t)
    return t.detach().to(device="cuda", dtype=torch.float32).contiguous()


def next_state(env, obs_deque, obs, reset):
    """Frame stack of the last ``maxlen`` prepped observations, oldest first on axis 0.
    On ``reset`` the passed ``obs`` is dropped and ``env.reset()`` is stacked behind
    ``maxlen-1`` zero frames (float64 result, like the reference)."""
    if reset:
        obs = env.reset()
        for _ in range(obs_deque.maxlen - 1):
            obs_deque.append(np.zeros(obs.shape))
    obs_deque.append(obs)
    return np.concatenate(obs_deque, axis=0)


def sample_action(pi, rand_nums=None):
    """Inverse-CDF sampling from probability vectors ``pi`` (..., A).  Returns a float tensor
    of shape ``pi.shape[:-1]`` holding the first index whose running fp32 cumsum is >= the
    uniform, or -1 if none is.  ``rand_nums`` (same leading shape) replaces the uniforms the
    reference draws with ``torch.rand``; without it they come from torch's device generator."""
    p = _dev(pi)
    lead = p.shape[:-1]
    if rand_nums is None:
        u = torch.rand(lead, device=p.device, dtype=torch.float32)
    else:
        u = _dev(rand_nums).reshape(lead)
    out = torch.empty(lead, device=p.device, dtype=torch.float32)
    if out.numel():
        ops.sample_probs(p, u.contiguous(), out)
    return out


def discount(array, dones, discount_factor, n_tsteps=None):
    """Reverse discounted sum with resets: ``y[i] = x[i] + g*(0 if dones[i]==1 else y[i+1])``.

    Bit-identical to the reference's sequential loop.  ``n_tsteps`` (optional) declares that
    the array is rows of that length each ending in ``dones == 1`` (what Runner produces), which
    lets the rows be scanned in parallel; this is verified on the device and, if it does not hold,
    the array is re-scanned there as one row (the reference's flat semantics).
    Without it the array is scanned as one row.
    """
    x, d = _dev(array).reshape(-1), _dev(dones).reshape(-1)
    n = x.numel()
    if d.numel() != n:
        raise ValueError("array and dones must have the same length")
    if n_tsteps is None or n == 0:
        return ops.discount_rows(x, d, discount_factor, 1, n)
    if n % n_tsteps:
        raise ValueError("len(array) is not a multiple of n_tsteps")
    err = torch.zeros(1, dtype=torch.int32, device=x.device)
    return ops.discount_rows(x, d, discount_factor, n // n_tsteps, n_tsteps, err=err)
