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
                 value_activation="tanh"):
        self.in_channels, self.policy_channels = in_channels, policy_channels
        self.width, self.num_blocks, self.recall = width, num_blocks, recall
        self.value_activation = value_activation

    @classmethod
    def from_state_dict(cls, sd, value_activation="tanh"):
        proj = sd["projection.0.weight"]
        width, in_channels = int(proj.shape[0]), int(proj.shape[1])
        recall = "recur_module.0.weight" in sd
        n_block_convs = sum(1 for k in sd if ".before_shortcut." in k)
        policy_channels = int(sd["policy_head.layers.2.weight"].shape[0])
        return cls(in_channels, policy_channels, width, n_block_convs // 2, recall, value_activation)


class Network_Manager:
    """Same public surface as the reference's class.  `model` is either a torch
    module with the reference's parameter names (its ``state_dict()`` is read) or
    a dict name -> array; `model.recurrent` must be True for the fused kernel."""

    def __init__(self, model, value_activation="tanh"):
        self.model = model
        if isinstance(model, dict):
            self._sd = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                        for k, v in model.items()}
            self.recurrent = True
        else:
            if not hasattr(model, "recurrent") or not isinstance(model.recurrent, bool):
                raise Exception('You need to add a "recurrent" boolean attribute to the model')
            if not model.recurrent:
                raise NotImplementedError("only the recurrent square-conv net runs on the fused kernel")
            self._sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
            self.recurrent = True
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
                                     value_activation=s.value_activation, recurrent_iterations=iters_to_do)
        logits, value, _ = self._engine.net_forward(state, want_probs=False)
        b = logits.shape[0]
        return logits.reshape(b, self._spec.policy_channels, 3, 3), value.reshape(b, 1)
