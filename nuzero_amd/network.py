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
                 value_activation="tanh", arch="recurrent", kernel_size=3):
        self.in_channels, self.policy_channels = in_channels, policy_channels
        self.width, self.num_blocks, self.recall = width, num_blocks, recall
        self.value_activation = value_activation
        self.arch, self.kernel_size = arch, kernel_size      # "recurrent" | "resnet" | "convnet"

    @classmethod
    def from_state_dict(cls, sd, value_activation="tanh"):
        """Recognise RecurrentNet / ResNet / ConvNet (hex=False) by their parameter names."""
        policy_channels = int(sd["policy_head.layers.2.weight"].shape[0])
        n_block_convs = sum(1 for k in sd if ".before_shortcut." in k)
        if "projection.0.weight" in sd:
            first = sd["projection.0.weight"]
            return cls(int(first.shape[1]), policy_channels, int(first.shape[0]), n_block_convs // 2,
                       "recur_module.0.weight" in sd, value_activation, "recurrent", 3)
        if "input_block.0.weight" in sd:
            first = sd["input_block.0.weight"]
            return cls(int(first.shape[1]), policy_channels, int(first.shape[0]), n_block_convs // 2, False,
                       value_activation, "resnet", 3)
        if "general_module.0.weight" in sd:
            first = sd["general_module.0.weight"]
            n_layers = sum(1 for k in sd if k.startswith("general_module.")) - 1
            return cls(int(first.shape[1]), policy_channels, int(first.shape[0]), n_layers, False,
                       value_activation, "convnet", int(first.shape[2]))
        raise ValueError("unrecognised network: expected RecurrentNet, ResNet or ConvNet parameter names")


class Network_Manager:
    """Same public surface as the reference's class.  `model` is either a torch
    module with the reference's parameter names (its ``state_dict()`` is read) or
    a dict name -> array; `model.recurrent` must be True for the fused kernel."""

    def __init__(self, model, value_activation="tanh"):
        self.model = model
        if isinstance(model, dict):
            self._sd = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                        for k, v in model.items()}
            self.recurrent = "projection.0.weight" in self._sd
        else:
            if not hasattr(model, "recurrent") or not isinstance(model.recurrent, bool):
                raise Exception('You need to add a "recurrent" boolean attribute to the model')
            self._sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
            self.recurrent = model.recurrent
        self._spec = RecurrentNetSpec.from_state_dict(self._sd, value_activation)
        self._engine = None

    def is_recurrent(self):
        return self.recurrent

    def get_model(self):
        return self.model

    def check_devices(self):
        """The reference moves the module to cuda here (Network_Manager.py:41-44);
        the engine copies the weights itself, so nothing to do."""

    def state_dict(self):
        return self._sd

    def spec(self):
        return self._spec

    def inference(self, state, training, iters_to_do=2, interim_thought=None):
        """(policy_logits [B,P,3,3], value [B,1]) on the GPU through the same MFMA
        kernel the search uses (Network_Manager.py:46-64, eval/no_grad branch)."""
        if training:
            raise NotImplementedError("the engine only evaluates; training stays in PyTorch")
        from .engine import SelfPlayEngine
        from .search_config import legacy_ttt_search_config
        if self._engine is None or self._engine.net_spec["iters"] != iters_to_do:
            if self._engine is None:
                self._engine = SelfPlayEngine(legacy_ttt_search_config(), 16)
            s = self._spec
            self._engine.set_weights(self._sd, width=s.width, num_blocks=s.num_blocks, recall=s.recall,
                                     value_activation=s.value_activation, recurrent_iterations=iters_to_do,
                                     arch=s.arch, kernel_size=s.kernel_size)
        logits, value, _ = self._engine.net_forward(state, want_probs=False)
        b = logits.shape[0]
        return logits.reshape(b, self._spec.policy_channels, 3, 3), value.reshape(b, 1)
