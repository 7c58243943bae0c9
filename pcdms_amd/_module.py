"""The slice of the ``torch.nn.Module`` surface the reference's drivers / pipelines touch on their model objects
(``.eval()``, ``.half()``, ``.cuda()``, ``.requires_grad_(False)``, ``.parameters()``, ``.modules()`` ...), for classes whose
weights live as packed bf16 device buffers rather than ``nn.Parameter``s.  Inference only: ``train(True)`` raises."""
from __future__ import annotations

import torch


class ModuleSurface:
    training = False

    def eval(self):
        return self

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("pcdms_amd models are inference-only (training is out of scope: SURVEY.md §2 rows 6, 14)")
        return self

    def requires_grad_(self, requires_grad: bool = False):
        if requires_grad:
            raise NotImplementedError("pcdms_amd models are inference-only")
        return self

    def half(self):
        return self.to(torch.float16)

    def float(self):
        return self.to(torch.float32)

    def cuda(self, device=None):
        return self.to(torch.device("cuda", torch.cuda.current_device() if device is None else device))

    def modules(self):
        yield self

    def named_parameters(self):
        """(name, fp32 host tensor) of the loaded state dict -- what ``sum(p.numel() for p in m.parameters())`` needs."""
        yield from (self.state_dict() or {}).items()

    def parameters(self):
        for _, v in self.named_parameters():
            yield v
