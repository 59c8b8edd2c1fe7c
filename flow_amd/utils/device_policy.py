"""DevicePolicy: a torch fully connected Gaussian policy in the form libflowsim's policy-in-the-loop entry points take.

What the reference's rollout workers evaluate per step through RLlib (examples/train.py:149-160: the default model,
``fcnet_hiddens [32, 32, 32]``, tanh, a diagonal Gaussian over the action) is here a flat float32 weight buffer on the
device that ``fs_policy_rollout_dev`` / ``fs_policy_act_dev`` read (include/flowsim.h ``fs_policy``): K steps of
policy -> action -> Env.step are ONE kernel launch, observations and actions never leave the chip's registers.
"""
import ctypes as C

from flow_amd import _lib as L


class DevicePolicy(object):
    """``hidden``: the ``nn.Linear`` layers of the trunk (1..3, 32 units each, tanh between them); ``head``: the output
    ``nn.Linear`` (2 outputs = mean and log std; 1 output with ``log_std`` a 1-element parameter / tensor).  ``sync()``
    copies the current parameter values into the packed device buffer (call it after every optimiser step)."""

    def __init__(self, hidden, head, log_std=None, seed=0):
        import torch
        self.torch = torch
        self.hidden, self.head, self.log_std_param = list(hidden), head, log_std
        if not 1 <= len(self.hidden) <= 3 or any(l.out_features != 32 for l in self.hidden):
            raise NotImplementedError("DevicePolicy: 1..3 hidden layers of 32 units (the kernel's model class)")
        n_out = 1 if log_std is not None else 2
        if head.out_features != n_out or head.in_features != 32:
            raise NotImplementedError("DevicePolicy: the head maps 32 units to %d output(s)" % n_out)
        dev = head.weight.device
        n = sum(l.weight.numel() + l.bias.numel() for l in self.hidden) + head.weight.numel() + head.bias.numel()
        self.buf = torch.zeros(n, dtype=torch.float32, device=dev)
        self.ls = torch.zeros(1, dtype=torch.float32, device=dev) if log_std is not None else None
        self.struct = L.fs_policy(struct_size=C.sizeof(L.fs_policy), obs_dim=self.hidden[0].in_features,
                                  num_hidden=len(self.hidden), hidden_width=32, activation=0,
                                  weights_dev=self.buf.data_ptr(),
                                  log_std_dev=self.ls.data_ptr() if self.ls is not None else None, seed=int(seed))
        self.sync()

    def sync(self):
        torch = self.torch
        with torch.no_grad():
            parts = []
            for l in self.hidden + [self.head]:
                parts += [l.weight.reshape(-1), l.bias.reshape(-1)]
            self.buf.copy_(torch.cat([p.detach().float() for p in parts]))
            if self.ls is not None:
                self.ls.copy_(self.log_std_param.detach().float().reshape(1))

    def reference(self, obs):
        """(mean, log std) of the same network evaluated by torch (float32): agrees with the kernels to ~1e-6."""
        torch = self.torch
        h = obs
        for l in self.hidden:
            h = torch.tanh(l(h))
        out = self.head(h)
        if self.ls is not None:
            return out[..., 0], self.log_std_param.reshape(1).expand_as(out[..., 0])
        return out[..., 0], out[..., 1]
