"""Value-net definition with the reference's parameter naming, so checkpoints / TorchScript files interchange.

The architecture contract is the reference's Net2 (/root/reference/cfvpy/models.py:64-94 with build_mlp :20-53):
    n_layers x [Linear -> LayerNorm? -> GELU -> Dropout?] -> Linear(output), output layer scaled by 0.01 at init.
The reference lays each hidden layer out as FOUR consecutive nn.Sequential slots (linear, norm, act, dropout; empty
nn.Sequential() where a slot is unused), which is what makes the state_dict keys `body.0.*, body.1.*, body.4.*,
body.5.*, output.*` -- rebel_amd.rela.ModelLocker and librebel_hip's rbl_engine_set_net_mlp read exactly those keys.
This file restates that contract; it is used by the tests / bench to make random-init nets of the right shape and by
users who train with the unmodified reference trainer (which brings its own class -- both produce the same keys).
"""
import torch
from torch import nn


def input_size(num_faces, num_dice):  # query width: player, traverser, one-hot last bid, two belief vectors
    return 2 + (2 * num_faces * num_dice + 1) + 2 * output_size(num_faces, num_dice)


def output_size(num_faces, num_dice):
    return num_faces ** num_dice


class GELU(nn.Module):
    def forward(self, x):
        return nn.functional.gelu(x)


class Net2(nn.Module):
    def __init__(self, *, num_faces, num_dice, n_hidden=256, use_layer_norm=False, dropout=0, n_layers=3):
        super().__init__()
        n_in = input_size(num_faces, num_dice)
        slots, width = [], n_in
        act = GELU()
        for _ in range(n_layers):
            slots += [nn.Linear(width, n_hidden),
                      nn.LayerNorm(n_hidden) if use_layer_norm else nn.Sequential(),
                      act,
                      nn.Dropout(dropout) if dropout > 0 else nn.Sequential()]
            width = n_hidden
        self.body = nn.Sequential(*slots)
        self.output = nn.Linear(width, output_size(num_faces, num_dice))
        with torch.no_grad():  # initial predictions close to zero
            self.output.weight.data *= 0.01
            self.output.bias *= 0.01

    def forward(self, packed_input: torch.Tensor):
        return self.output(self.body(packed_input))


def mlp_weights_from_state_dict(sd):
    """state_dict (reference key names) -> (layers, ln, w_out, b_out) as numpy, the form capi.Engine.set_net_mlp takes.

    Raises ValueError when the keys do not describe a Net2-shaped MLP."""
    import re

    lin = sorted(int(m.group(1)) for k in sd for m in [re.fullmatch(r"body\.(\d+)\.weight", k)]
                 if m and sd[k].dim() == 2)
    if "output.weight" not in sd or "output.bias" not in sd:
        raise ValueError("not a Net2 state_dict: no output.weight/bias")
    layers, ln = [], []
    for i in lin:
        if i % 4 != 0:
            raise ValueError(f"unexpected Linear at body.{i} (Net2 puts them at multiples of 4)")
        layers.append((sd[f"body.{i}.weight"].detach().float().cpu().numpy(),
                       sd[f"body.{i}.bias"].detach().float().cpu().numpy()))
        if f"body.{i + 1}.weight" in sd:
            ln.append((sd[f"body.{i + 1}.weight"].detach().float().cpu().numpy(),
                       sd[f"body.{i + 1}.bias"].detach().float().cpu().numpy()))
    if ln and len(ln) != len(layers):
        raise ValueError("LayerNorm present on some hidden layers only")
    if not layers:
        raise ValueError("Net2 with n_layers=0 is not supported by the MFMA forward")
    return (layers, ln or None, sd["output.weight"].detach().float().cpu().numpy(),
            sd["output.bias"].detach().float().cpu().numpy())
