"""Network hand-off with the reference's Network_Manager surface
(Neural_Networks/Network_Manager.py:14-64).

The reference ships a whole torch module to every Gamer through Ray
(`shared_storage.get`, Training/Gamer.py:40,61).  Here the engine only needs the
weight tensors; this wrapper keeps the reference's constructor/`inference`
signature so trainer-side code can keep passing a Network_Manager around, and
adds `spec()` = what nz_engine_set_weights needs.
"""
import numpy as np
import torch


class RecurrentNetSpec:
    """RecurrentNet(in_channels, policy_channels, num_filters, num_blocks, recall,
    policy_head="conv", value_head="reduce", value_activation, hex=False)
    (Neural_Networks/Architectures/RecurrentNet.py:18-79) as plain numbers."""

    def __init__(self, in_channels=2, policy_channels=1, width=64, num_blocks=2, recall=True,
                 value_activation="tanh", arch="recurrent", kernel_size=3, hex=False):
        self.in_channels, self.policy_channels = in_channels, policy_channels
        self.width, self.num_blocks, self.recall = width, num_blocks, recall
        self.value_activation = value_activation
        self.arch, self.kernel_size = arch, kernel_size      # "recurrent" | "resnet" | "convnet"
        self.hex = hex                                        # every conv a hexagdly.Conv2d(kernel_size=1): board nets only

    @classmethod
    def from_state_dict(cls, sd, value_activation="tanh"):
        """Recognise RecurrentNet / ResNet / ConvNet by their parameter names; hex=True models hold
        `<layer>.kernel0` / `<layer>.kernel1` (hexagdly.Conv2d) where square ones hold `<layer>.weight`."""
        hexnet = "policy_head.layers.2.kernel0" in sd
        suffix = "kernel0" if hexnet else "weight"
        policy_channels = int(sd["policy_head.layers.2." + suffix].shape[0])
        n_block_convs = sum(1 for k in sd if ".before_shortcut." in k and k.endswith(suffix))
        if "projection.0." + suffix in sd:
            first = sd["projection.0." + suffix]
            return cls(int(first.shape[1]), policy_channels, int(first.shape[0]), n_block_convs // 2,
                       "recur_module.0." + suffix in sd, value_activation, "recurrent", 3, hexnet)
        if "input_block.0." + suffix in sd:
            first = sd["input_block.0." + suffix]
            return cls(int(first.shape[1]), policy_channels, int(first.shape[0]), n_block_convs // 2, False,
                       value_activation, "resnet", 3, hexnet)
        if "general_module.0." + suffix in sd:
            first = sd["general_module.0." + suffix]
            n_layers = sum(1 for k in sd if k.startswith("general_module.") and k.endswith(suffix)) - 1
            return cls(int(first.shape[1]), policy_channels, int(first.shape[0]), n_layers, False,
                       value_activation, "convnet", 1 if hexnet else int(first.shape[2]), hexnet)
        raise ValueError("unrecognised network: expected RecurrentNet, ResNet or ConvNet parameter names")


class Network_Manager:
    """Same public surface as the reference's class.  `model` is either a torch
    module with the reference's parameter names (its ``state_dict()`` is read) or
    a dict name -> array; `model.recurrent` must be True for the fused kernel."""

    def __init__(self, model, value_activation="tanh"):
        self.model = model
        self._value_activation = value_activation
        if not isinstance(model, dict):
            if not hasattr(model, "recurrent") or not isinstance(model.recurrent, bool):
                raise Exception('You need to add a "recurrent" boolean attribute to the model')
        self.version = 0            # bumped whenever the weights the engine should use have changed
        self._stamp = None
        self._engine = None
        self._engine_version = -1
        self.refresh()

    # ---- weight hand-off -------------------------------------------------------------------------
    # The reference's trainer keeps ONE long-lived Network_Manager / module and trains it in place
    # (Training/AlphaZero.py:152,293,462); Ray's get() handed every Gamer a fresh copy of it per game
    # (Training/Gamer.py:37,61).  Without Ray the object's identity never changes, so the weights are
    # re-read whenever the module's parameters were written to: every in-place write (optimizer step,
    # load_state_dict) bumps torch's per-tensor version counter.
    def _current_stamp(self):
        if isinstance(self.model, dict):
            return None
        return tuple((id(t), t._version) for t in self.model.state_dict(keep_vars=True).values())

    def refresh(self):
        """Snapshot the model's weights now (always) and bump `version`."""
        model = self.model
        if isinstance(model, dict):
            self._sd = {k: (v.detach().cpu().numpy().copy() if isinstance(v, torch.Tensor) else np.array(v))
                        for k, v in model.items()}
            self.recurrent = "projection.0.weight" in self._sd or "projection.0.kernel0" in self._sd
        else:
            self._sd = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
            self.recurrent = model.recurrent
        self._stamp = self._current_stamp()
        self._spec = RecurrentNetSpec.from_state_dict(self._sd, self._value_activation)
        self.version += 1
        return self.version

    def sync(self):
        """Re-read the weights if the module was trained / loaded since the last snapshot.  Returns `version`.
        (dict models are plain data: call refresh() after editing one in place.)"""
        if self._stamp is not None and self._current_stamp() != self._stamp:
            self.refresh()
        return self.version

    def is_recurrent(self):
        return self.recurrent

    def get_model(self):
        return self.model

    def check_devices(self):
        """The reference moves the module to cuda here (Network_Manager.py:41-44);
        the engine copies the weights itself, so nothing to do."""

    def model_to_cpu(self):
        """Network_Manager.py:32-33; the engine holds its own copy of the weights, nothing moves."""

    def model_to_device(self):
        """Network_Manager.py:35-36; see model_to_cpu."""

    def state_dict(self):
        self.sync()
        return self._sd

    def spec(self):
        self.sync()
        return self._spec

    def inference(self, state, training, iters_to_do=2, interim_thought=None):
        """(policy_logits [B,P,3,3], value [B,1]) on the GPU through the same MFMA
        kernel the search uses (Network_Manager.py:46-64, eval/no_grad branch)."""
        if training:
            raise NotImplementedError("the engine only evaluates; training stays in PyTorch")
        version = self.sync()
        s = self._spec
        rows, cols = int(state.shape[-2]), int(state.shape[-1])
        if (s.in_channels, s.policy_channels, rows, cols, s.hex) != (2, 1, 3, 3, False):
            return self._board_inference(state, iters_to_do, rows, cols)
        from .engine import SelfPlayEngine
        from .search_config import legacy_ttt_search_config
        if self._engine is None or self._engine.net_spec["iters"] != iters_to_do or self._engine_version != version:
            if self._engine is None:
                self._engine = SelfPlayEngine(legacy_ttt_search_config(), 16)
            self._engine_version = version
            self._engine.set_weights(self._sd, width=s.width, num_blocks=s.num_blocks, recall=s.recall,
                                     value_activation=s.value_activation, recurrent_iterations=iters_to_do,
                                     arch=s.arch, kernel_size=s.kernel_size)
        logits, value, _ = self._engine.net_forward(state, want_probs=False)
        b = logits.shape[0]
        return logits.reshape(b, self._spec.policy_channels, 3, 3), value.reshape(b, 1)

    def board_net(self, rows, cols, max_batch, recurrent_iterations=1, device=0):
        """The model as a nuzero_amd.boardnet.BoardNet for rows x cols boards (SCS): the network the
        SCS self-play path evaluates on the device."""
        from .boardnet import BoardNet
        s = self._spec
        net = BoardNet(s.arch, s.in_channels, s.policy_channels, rows, cols, width=s.width, num_blocks=s.num_blocks,
                       recall=s.recall, value_activation=s.value_activation, kernel_size=s.kernel_size,
                       max_batch=max_batch, device=device, hex=s.hex)
        net.set_weights(self._sd, recurrent_iterations or 1)
        return net

    def _board_inference(self, state, iters_to_do, rows, cols):
        x = torch.as_tensor(state, dtype=torch.float32)
        key = (rows, cols, iters_to_do, max(int(x.shape[0]), 16), self.sync())
        if getattr(self, "_board", None) is None or self._board[0] != key:
            if getattr(self, "_board", None) is not None:
                self._board[1].close()
            self._board = (key, self.board_net(rows, cols, key[3], iters_to_do))
        net = self._board[1]
        x = x.to(net.device).contiguous()
        _, value, logits = net.forward(x, want_logits=True)
        b = x.shape[0]
        return logits.reshape(b, self._spec.policy_channels, rows, cols), value.reshape(b, 1)
