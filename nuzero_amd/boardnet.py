"""Policy/value network on boards of any size, on the GPU (C ABI nz_boardnet_*).

The square-conv (hex=False) RecurrentNet / ResNet / ConvNet of the reference
(Neural_Networks/Architectures/*.py) as hand-written implicit-GEMM FP32 MFMA kernels
(nuzero_amd/csrc/boardnet.hip).  `BoardNet.evaluator()` plugs into ScsSelfPlay.play in place of
a PyTorch model: same inputs (state images of the wave's leaves), same outputs (softmax over all
logits, value), no PyTorch in the evaluation.
"""
import ctypes
from ctypes import byref, c_void_p

import numpy as np
import torch

from . import _lib
from ._lib import lib

_ARCH = {"recurrent": _lib.NZ_ARCH_RECURRENT, "resnet": _lib.NZ_ARCH_RESNET, "convnet": _lib.NZ_ARCH_CONVNET}


class BoardNet:
    def __init__(self, arch, in_channels, policy_channels, rows, cols, width=64, num_blocks=2, recall=True,
                 value_activation="tanh", kernel_size=3, max_batch=1024, device=0, hex=False):
        if not torch.cuda.is_available():
            raise RuntimeError("nuzero_amd needs a ROCm GPU; there is no CPU fallback")
        self.device = torch.device("cuda", device)
        self.arch, self.rows, self.cols = arch, int(rows), int(cols)
        self.in_channels, self.policy_channels = int(in_channels), int(policy_channels)
        self.num_actions = self.policy_channels * self.rows * self.cols
        self.max_batch = int(max_batch)
        self.desc = _lib.NetDesc(in_channels, policy_channels, width, num_blocks, int(bool(recall)),
                                 _lib.NZ_ACT_RELU if value_activation == "relu" else _lib.NZ_ACT_TANH,
                                 _ARCH[arch], 1 if hex else kernel_size, int(bool(hex)))
        self._h = c_void_p(0)
        st = lib.nz_boardnet_create(byref(self._h), byref(self.desc), self.rows, self.cols, self.max_batch, device)
        if st != _lib.NZ_OK:
            raise _lib.NzError(st, (lib.nz_boardnet_last_error(None) or b"").decode())

    def _check(self, st):
        if st != _lib.NZ_OK:
            raise _lib.NzError(st, (lib.nz_boardnet_last_error(self._h) or b"").decode())

    def close(self):
        if self._h.value:
            lib.nz_boardnet_destroy(self._h)
            self._h = c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_weights(self, weights, recurrent_iterations=1):
        """weights: the reference's state_dict (name -> array / tensor), in state_dict order."""
        tensors = [torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v).detach().float().contiguous().cpu()
                   for v in weights.values()]
        ptrs = (c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
        self._check(lib.nz_boardnet_set_weights(self._h, ptrs, len(tensors), int(recurrent_iterations)))

    def fused(self, enable=None):
        """Whether the one-launch form of the network (all layers in LDS) exists for this net / board / max_batch;
        `enable` = True / False switches its use on or off (both forms give the same floats)."""
        avail = ctypes.c_int32(0)
        self._check(lib.nz_boardnet_fused(self._h, -1 if enable is None else int(bool(enable)), byref(avail)))
        return bool(avail.value)

    @property
    def flops_per_position(self):
        return int(lib.nz_boardnet_flops(self._h))

    def forward(self, images, n_dev=None, want_logits=False):
        """images: float32 device tensor [n, C, rows, cols] -> (probs [n, A], value [n]) (+ logits)."""
        assert images.is_cuda and images.dtype == torch.float32 and images.is_contiguous()
        n = images.shape[0]
        assert tuple(images.shape[1:]) == (self.in_channels, self.rows, self.cols)
        probs = torch.empty((n, self.num_actions), dtype=torch.float32, device=self.device)
        value = torch.empty((n,), dtype=torch.float32, device=self.device)
        logits = torch.empty_like(probs) if want_logits else None
        self._check(lib.nz_boardnet_forward(
            self._h, c_void_p(images.data_ptr()), n, c_void_p(n_dev.data_ptr()) if n_dev is not None else None,
            c_void_p(logits.data_ptr()) if want_logits else None, c_void_p(probs.data_ptr()),
            c_void_p(value.data_ptr()), c_void_p(torch.cuda.current_stream().cuda_stream)))
        return (probs, value, logits) if want_logits else (probs, value)

    def evaluator(self, fixed_batch=None):
        """For ScsSelfPlay.play: images -> (probs, values).  With `fixed_batch` every call is launched for that many
        positions with the live count in device memory -- the launch shapes (and with them the kernel chosen per
        layer) of the in-library move loop, nz_scs_search_play, at fixed_batch = its number of games."""
        if fixed_batch is None:
            return lambda images: self.forward(images.contiguous())
        buf = torch.zeros((int(fixed_batch), self.in_channels, self.rows, self.cols), dtype=torch.float32,
                          device=self.device)
        n_dev = torch.zeros((1,), dtype=torch.int32, device=self.device)

        def ev(images):
            n = images.shape[0]
            buf[:n].copy_(images)
            n_dev.fill_(n)
            probs, value = self.forward(buf, n_dev=n_dev)
            return probs[:n], value[:n]
        return ev
