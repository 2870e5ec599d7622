"""CPU restatement of the reference's gated feed-forward activation (TEST INFRASTRUCTURE ONLY).

`FlashT5DenseGatedAct.forward` (reference src/model/modeling_flash_t5.py:139-142):
    hidden_act = self.act(self.wi_0(hidden_states)); hidden_linear = self.wi_1(hidden_states); hidden_states = hidden_act * hidden_linear
with `self.act = torch.nn.GELU(approximate='tanh') if config.use_gelu_act else torch.nn.ReLU()` (:134).
The tanh GELU is written out (the published formula torch implements) so that the oracle does not lean on the op it checks;
tests/test_oracle_golden.py pins it against torch.nn.functional.gelu(approximate='tanh') itself.  fp64 math on the given values."""
import math

import torch

K = math.sqrt(2.0 / math.pi)
C = 0.044715


def _act(x, act):
    if act == "relu":
        return torch.clamp(x, min=0.0), (x > 0).to(x.dtype)
    if act != "gelu_tanh":
        raise ValueError(act)
    u = K * (x + C * x ** 3)
    t = torch.tanh(u)
    return 0.5 * x * (1 + t), 0.5 * (1 + t) + 0.5 * x * (1 - t * t) * K * (1 + 3 * C * x * x)


def gated_act_oracle(h0, h1, act="gelu_tanh"):
    """act(h0) * h1 in fp64 (the reference rounds act(h0) to the tensor dtype before the multiply: half an ulp the comparison's
    tolerance covers)."""
    a, _ = _act(h0.double(), act)
    return a * h1.double()


def gated_act_bwd_oracle(dout, h0, h1, act="gelu_tanh"):
    """(dh0, dh1) = (dout * h1 * act'(h0), dout * act(h0)) in fp64"""
    a, da = _act(h0.double(), act)
    g = dout.double()
    return g * h1.double() * da, g * a
