"""Oracle: square-conv RecurrentNet forward in plain PyTorch fp32 (test
infrastructure only).

Restates /root/reference/Neural_Networks/Architectures/RecurrentNet.py:18-99
with ``hex=False`` and the blocks it uses (blocks.py:12-41 BasicBlock,
blocks.py:46-92 Reduce_ValueHead, blocks.py:130-170 Reduce_PolicyHead), plus the
inference wrapper of Neural_Networks/Network_Manager.py:46-64.

All convolutions are 3x3, stride 1, zero 'same' padding, bias-free.  Weights are
a dict keyed by the reference's ``state_dict`` names, so a reference checkpoint
(``model_state_dict``) loads unchanged.
"""
import numpy as np
import torch
import torch.nn.functional as F


def head_channels(width, out_channels, n_layers):
    """Channel schedule shared by both heads: float steps truncated by int()
    (blocks.py:56-66,144-153).  width 64 -> policy [64,32,1], value [64,48,32,16,1]."""
    step = (out_channels - width) / n_layers
    chans = [width]
    prev = float(width)
    for _ in range(n_layers):
        prev = prev + step
        chans.append(int(prev))
    return chans


def param_shapes(in_channels, policy_channels, width, num_blocks, recall=True):
    """Ordered (name, shape) list; names as in the reference's state_dict."""
    out = [("projection.0.weight", (width, in_channels, 3, 3))]
    idx = 0
    if recall:
        out.append(("recur_module.0.weight", (width, width + in_channels, 3, 3)))
        idx = 1
    for b in range(num_blocks):
        out.append((f"recur_module.{idx + b}.before_shortcut.0.weight", (width, width, 3, 3)))
        out.append((f"recur_module.{idx + b}.before_shortcut.2.weight", (width, width, 3, 3)))
    pc = head_channels(width, policy_channels, 2)
    for i in range(2):
        out.append((f"policy_head.layers.{2 * i}.weight", (pc[i + 1], pc[i], 3, 3)))
    vc = head_channels(width, 1, 4)
    for i in range(4):
        out.append((f"value_head.layers.{2 * i}.weight", (vc[i + 1], vc[i], 3, 3)))
    return out


class RecurrentNetRef:
    def __init__(self, weights, in_channels, policy_channels, width=64, num_blocks=2,
                 recall=True, value_activation="tanh"):
        self.w = {k: torch.as_tensor(np.asarray(v), dtype=torch.float32) for k, v in weights.items()}
        self.in_channels = in_channels
        self.policy_channels = policy_channels
        self.width = width
        self.num_blocks = num_blocks
        self.recall = recall
        self.value_activation = value_activation
        self.recurrent = True

    def _conv(self, x, name):
        return F.conv2d(x, self.w[name], None, 1, "same")

    def forward(self, x, iters):
        """RecurrentNet.py:82-99."""
        x = torch.as_tensor(x, dtype=torch.float32)
        thought = F.relu(self._conv(x, "projection.0.weight"))
        base = 1 if self.recall else 0
        for _ in range(iters):
            if self.recall:
                thought = torch.cat([thought, x], 1)           # RecurrentNet.py:91
                thought = self._conv(thought, "recur_module.0.weight")   # no activation
            for b in range(self.num_blocks):                     # blocks.py:37-41
                pre = f"recur_module.{base + b}.before_shortcut."
                y = F.relu(self._conv(thought, pre + "0.weight"))
                y = self._conv(y, pre + "2.weight")
                thought = F.relu(y + thought)
        p = F.relu(self._conv(thought, "policy_head.layers.0.weight"))
        p = self._conv(p, "policy_head.layers.2.weight")
        act = torch.tanh if self.value_activation == "tanh" else F.relu
        v = thought
        for i in range(4):
            v = self._conv(v, f"value_head.layers.{2 * i}.weight")
            if i != 3:
                v = act(v)
        v = torch.tanh(v.mean(dim=(1, 2, 3)).reshape(-1, 1))     # blocks.py:82-84
        return p, v

    def inference(self, state, iters):
        """Network_Manager.py:46-64 (eval mode, no_grad); numpy outputs.

        Runs with one intra-op thread: the golden vectors were made that way
        (the reference's batch-1 inference is fastest single-threaded, SURVEY.md
        section 6) and oneDNN's conv result depends on the thread count in the
        last bit."""
        n = torch.get_num_threads()
        torch.set_num_threads(1)
        try:
            with torch.no_grad():
                p, v = self.forward(state, iters)
        finally:
            torch.set_num_threads(n)
        return p.numpy(), v.numpy()


def _heads(w, trunk, value_activation):
    """Reduce_PolicyHead + Reduce_ValueHead (blocks.py:46-92,130-170) on the trunk output."""
    conv = lambda x, name: F.conv2d(x, w[name], None, 1, "same")
    p = F.relu(conv(trunk, "policy_head.layers.0.weight"))
    p = conv(p, "policy_head.layers.2.weight")
    act = torch.tanh if value_activation == "tanh" else F.relu
    v = trunk
    for i in range(4):
        v = conv(v, f"value_head.layers.{2 * i}.weight")
        if i != 3:
            v = act(v)
    return p, torch.tanh(v.mean(dim=(1, 2, 3)).reshape(-1, 1))


class FeedForwardRef:
    """ResNet / ConvNet with hex=False (Architectures/ResNet.py:64-70, ConvNet.py:52-57)."""

    def __init__(self, weights, arch, num_blocks, value_activation="tanh"):
        self.w = {k: torch.as_tensor(np.asarray(v), dtype=torch.float32) for k, v in weights.items()}
        self.arch, self.num_blocks, self.value_activation = arch, num_blocks, value_activation
        self.recurrent = False

    def forward(self, x):
        x = torch.as_tensor(x, dtype=torch.float32)
        conv = lambda t, name: F.conv2d(t, self.w[name], None, 1, "same")
        if self.arch == "resnet":
            t = F.relu(conv(x, "input_block.0.weight"))
            for b in range(self.num_blocks):
                pre = f"residual_blocks.{b}.before_shortcut."
                y = conv(F.relu(conv(t, pre + "0.weight")), pre + "2.weight")
                t = F.relu(y + t)
        else:
            t = F.elu(conv(x, "general_module.0.weight"))
            for i in range(self.num_blocks):
                t = F.elu(conv(t, f"general_module.{2 * (i + 1)}.weight"))
        return _heads(self.w, t, self.value_activation)

    def inference(self, state, iters=None):
        n = torch.get_num_threads()
        torch.set_num_threads(1)
        try:
            with torch.no_grad():
                p, v = self.forward(state)
        finally:
            torch.set_num_threads(n)
        return p.numpy(), v.numpy()


# ---- hexagonal convolutions: PARITY UNPINNED ------------------------------------------------------------
# hexagdly is not installed in the build container, so nothing below could be checked against it or against
# outputs of the reference's hex=True nets.  It restates hexagdly.Conv2d(kernel_size=1, bias=False) from the
# package's documentation: a 7-cell neighbourhood on a grid whose odd columns sit half a cell lower -- the
# adjacency SCS_Game uses (Games/SCS/SCS_Game.py:1199-1243; tests/test_hex_oracle.py checks that much) --
# with kernel0 [out, in, 3, 1] = (N, centre, S) of the cell's own column and kernel1 [out, in, 2, 2] =
# [upper, lower] x [left column, right column].
def hex_conv2d(x, kernel0, kernel1):
    x = torch.as_tensor(x, dtype=torch.float32)
    k0 = torch.as_tensor(np.asarray(kernel0), dtype=torch.float32)
    k1 = torch.as_tensor(np.asarray(kernel1), dtype=torch.float32)
    n, _, rows, cols = x.shape
    out = F.conv2d(x, k0, None, 1, (1, 0))                           # own column: rows r-1, r, r+1
    xp = F.pad(x, (1, 1, 1, 1))                                      # zero border: xp[r + 1, c + 1] = x[r, c]
    even = (torch.arange(cols) % 2 == 0).reshape(1, 1, 1, cols)

    def shifted(dr, dc):                                             # t[r, c] = x[r + dr, c + dc] (0 outside)
        return xp[:, :, 1 + dr:1 + dr + rows, 1 + dc:1 + dc + cols]

    for side, dc in ((0, -1), (1, 1)):
        upper = torch.where(even, shifted(-1, dc), shifted(0, dc))   # even column: rows r-1 / r; odd: r / r+1
        lower = torch.where(even, shifted(0, dc), shifted(1, dc))
        out = out + F.conv2d(upper, k1[:, :, 0:1, side:side + 1]) + F.conv2d(lower, k1[:, :, 1:2, side:side + 1])
    return out


class HexNetRef:
    """RecurrentNet / ResNet / ConvNet with hex=True (every conv a hexagdly.Conv2d(kernel_size=1)); weights keyed
    `<layer>.kernel0` / `<layer>.kernel1` in state_dict order.  Parity unpinned (see above)."""

    def __init__(self, weights, arch, num_blocks, recall=True, value_activation="tanh"):
        self.w = {k: torch.as_tensor(np.asarray(v), dtype=torch.float32) for k, v in weights.items()}
        self.arch, self.num_blocks, self.recall, self.value_activation = arch, num_blocks, recall, value_activation
        self.recurrent = arch == "recurrent"

    def _conv(self, x, layer):
        return hex_conv2d(x, self.w[layer + ".kernel0"], self.w[layer + ".kernel1"])

    def forward(self, x, iters=1):
        x = torch.as_tensor(x, dtype=torch.float32)
        c = self._conv
        if self.arch == "recurrent":
            t = F.relu(c(x, "projection.0"))
            base = 1 if self.recall else 0
            for _ in range(iters):
                if self.recall:
                    t = c(torch.cat([t, x], 1), "recur_module.0")
                for b in range(self.num_blocks):
                    pre = f"recur_module.{base + b}.before_shortcut."
                    t = F.relu(c(F.relu(c(t, pre + "0")), pre + "2") + t)
        elif self.arch == "resnet":
            t = F.relu(c(x, "input_block.0"))
            for b in range(self.num_blocks):
                pre = f"residual_blocks.{b}.before_shortcut."
                t = F.relu(c(F.relu(c(t, pre + "0")), pre + "2") + t)
        else:
            t = F.elu(c(x, "general_module.0"))
            for i in range(self.num_blocks):
                t = F.elu(c(t, f"general_module.{2 * (i + 1)}"))
        p = c(F.relu(c(t, "policy_head.layers.0")), "policy_head.layers.2")
        act = torch.tanh if self.value_activation == "tanh" else F.relu
        v = t
        for i in range(4):
            v = c(v, f"value_head.layers.{2 * i}")
            if i != 3:
                v = act(v)
        return p, torch.tanh(v.mean(dim=(1, 2, 3)).reshape(-1, 1))

    def inference(self, state, iters=1):
        n = torch.get_num_threads()
        torch.set_num_threads(1)
        try:
            with torch.no_grad():
                p, v = self.forward(state, iters or 1)
        finally:
            torch.set_num_threads(n)
        return p.numpy(), v.numpy()
