"""Batched policy/value loss on the device (C ABI nz_loss_forward_backward, nuzero_amd/csrc/loss.hip).

`calculate_loss` has the arguments and the return value of AlphaZero.calculate_loss (Training/AlphaZero.py:891-921:
(value_loss, policy_loss, combined_loss)) but takes the targets as batch tensors -- what DeviceReplayBuffer hands out --
instead of a list of (value, policy list) tuples, and computes the whole batch in one launch instead of a Python loop
over the samples.  The losses are autograd-aware: the kernel also writes d(combined)/d(logits) and d(combined)/d(value),
so `combined_loss.backward()` works as in batch_update_weights (AlphaZero.py:884).
"""
from ctypes import c_void_p

import torch

from . import _lib
from ._lib import lib

POLICY_LOSSES = {"CEL": _lib.NZ_LOSS_CE, "KLD": _lib.NZ_LOSS_KLD, "MSE": _lib.NZ_LOSS_MSE}   # AlphaZero.py:325-333
VALUE_LOSSES = {"SE": _lib.NZ_LOSS_SE, "AE": _lib.NZ_LOSS_AE}                                # AlphaZero.py:335-339


class _FusedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, values, target_policies, target_values, policy_loss, value_loss, normalize_policy):
        if not logits.is_cuda:
            raise RuntimeError("nuzero_amd needs a ROCm GPU; there is no CPU fallback")
        B = logits.shape[0]
        x = logits.reshape(B, -1).float().contiguous()
        v = values.reshape(B).float().contiguous()
        tp = target_policies.reshape(B, -1).float().contiguous()
        tv = target_values.reshape(B).float().contiguous()
        assert tp.shape == x.shape
        if policy_loss == POLICY_LOSSES["MSE"] and not bool((tp != 0).any(dim=1).all()):
            # loss_functions.py:7-26 divides by the number of non-zero target entries of the sample
            raise ZeroDivisionError("masked MSE policy loss: a sample's target policy is all zeros")
        losses = torch.empty(3, dtype=torch.float32, device=x.device)
        dlogits = torch.empty_like(x)
        dvalues = torch.empty_like(v)
        work = torch.empty(2 * B, dtype=torch.float32, device=x.device)
        st = lib.nz_loss_forward_backward(
            c_void_p(x.data_ptr()), c_void_p(v.data_ptr()), c_void_p(tp.data_ptr()), c_void_p(tv.data_ptr()), B,
            x.shape[1], policy_loss, value_loss, int(bool(normalize_policy)), c_void_p(losses.data_ptr()),
            c_void_p(dlogits.data_ptr()), c_void_p(dvalues.data_ptr()), c_void_p(work.data_ptr()),
            c_void_p(torch.cuda.current_stream().cuda_stream))
        if st != _lib.NZ_OK:
            raise _lib.NzError(st, (lib.nz_loss_last_error() or b"").decode())
        ctx.save_for_backward(dlogits, dvalues)
        ctx.shapes = (logits.shape, values.shape)
        return losses[0], losses[1], losses[2]

    @staticmethod
    def backward(ctx, g_value, g_policy, g_combined):
        # The kernel differentiates the COMBINED loss (what the reference back-propagates, AlphaZero.py:882-884); its
        # two halves depend on disjoint inputs, so the parts' gradients are the same arrays.
        dlogits, dvalues = ctx.saved_tensors
        gl = dlogits * (g_combined + g_policy)
        gv = dvalues * (g_combined + g_value)
        return gl.reshape(ctx.shapes[0]), gv.reshape(ctx.shapes[1]), None, None, None, None, None


def calculate_loss(outputs, target_policies, target_values, policy_loss_function="CEL", value_loss_function="SE",
                   normalize_policy=False):
    """outputs = (policy_logits [B, P, H, W] or [B, A], values [B, 1] or [B]) as Network_Manager.inference returns them;
    target_policies [B, A]; target_values [B].  Returns (value_loss, policy_loss, combined_loss), 0-dim float32 tensors on
    the device.  Loss names are the reference's config strings (Learning.policy_loss / value_loss)."""
    logits, values = outputs
    return _FusedLoss.apply(logits, values, target_policies, target_values, POLICY_LOSSES[policy_loss_function],
                            VALUE_LOSSES[value_loss_function], normalize_policy)
