"""`Register` — the public façade, drop-in for ref:src/TorchRegister/torchregister.py:11-129.

Same constructor, `optim` and `__call__` signatures and the same criterion/weight branching
(ref:torchregister.py:70-106).  Extensions are keyword-only and default to reference behaviour:
optimizer ('sgd'|'adam'), honor_criterion, init, smooth_weight, flow_model ('unet' = the reference's
U-Net-generated flow, 'direct' = the flow field itself is the parameter); after `optim`, `.losses`
(the loss curve the reference only plots), `.final_theta` and `.best_idx` are available.
"""
import torch
from torch import cat

from .warpings import affine_register, flow_register, get_affine_warp, rigid_register


class Register():
    def __init__(self, mode='rigid', device='cpu', criterion=None, weight=None, grad_edges=False, debug=False, *,
                 optimizer='sgd', honor_criterion=False, init=None, smooth_weight=0.0, flow_model='unet'):
        '''
        Numerical registration on an AMD GPU (MI355X) behind the TorchRegister API.

        Parameters
        ----------
        mode : 'rigid', 'affine' or 'flow'. The default is 'rigid'.
        device : kept for signature compatibility; the tensors passed to optim()/__call__ must live on
            the GPU ('cuda'); CPU tensors raise (there is no CPU fallback).
        criterion : list of losses (nn.MSELoss, NCCLoss, SSDLoss are fused; anything else runs through
            the generic autograd path). For rigid/affine the reference ignores a user list:
            reproduced unless honor_criterion=True.
        weight : list of floats associated with criterion.
        grad_edges : must stay False (the reference's edge filter crashes when enabled).
        debug : print a one-line summary after optim.
        '''
        if mode not in ('rigid', 'affine', 'flow'):
            raise ValueError("mode must be 'rigid', 'affine' or 'flow'")
        self.criterion = criterion
        self.weight = weight
        self.mode = mode
        self.warp = None if mode == 'flow' else get_affine_warp
        self.device = device
        self.debug = debug
        self.theta = None
        self.grad_edges = grad_edges
        self.optimizer = optimizer
        self.honor_criterion = honor_criterion
        self.init = init
        self.smooth_weight = smooth_weight
        self.flow_model = flow_model
        self.losses = None
        self.final_theta = None
        self.best_idx = None

    def optim(self, moving, target, lr=1E-5, max_epochs=1000, n=32, per=0.1):
        '''
        Optimisation loop: moving, target [1,1,x,y(,z)] float32 GPU tensors (a leading batch > 1 of
        independent pairs is an extension).  Sets self.theta (best theta [B,nd,nd+1], or the flow
        [B,nd,...] of the last forward in flow mode) and self.warp.  Returns None.
        '''
        if self.mode == 'flow':
            kw = dict(mode='bilinear', n=n, lr=lr, max_epochs=max_epochs, optimizer=self.optimizer, smooth_weight=self.smooth_weight,
                      flow_model=self.flow_model)
            if self.criterion is not None and self.weight is not None:       # ref:torchregister.py:71-73
                kw.update(criterions=self.criterion, weights=self.weight)
            elif self.weight is not None:                                     # ref:torchregister.py:74-76
                kw.update(weights=self.weight)
            flowreg = flow_register(target.shape[2:], **kw).to(moving.device)
            flowreg.optimize(moving, target, self.device, self.debug)
            self.theta = flowreg.flow
            self.warp = flowreg.deform
            self.losses = flowreg.losses
            self.final_theta = flowreg.final_flow
            return

        fn = affine_register if self.mode == 'affine' else rigid_register
        info = {}
        kw = dict(lr=lr, epochs=max_epochs, per=per, device=self.device, debug=self.debug, grad_edges=self.grad_edges,
                  honor_criterion=self.honor_criterion, optimizer=self.optimizer, init=self.init, info=info)
        if self.criterion is not None and self.weight is not None:           # ref:torchregister.py:85-87,97-99
            kw.update(criterions=self.criterion, weights=self.weight)
        elif self.weight is not None:                                         # ref:torchregister.py:88-90,100-102
            kw.update(weights=self.weight)
        _, theta = fn(moving, target, **kw)
        self.theta = theta[-1]                                                # best theta (Q8)
        self.final_theta = theta[0]
        self.losses = info.get('losses')
        self.best_idx = info.get('best_idx')

    def __call__(self, moving):
        '''
        Warp moving [B,c,x,y(,z)] with the deformation found by optim: one fused launch for all
        channels (the reference loops over channels, ref:torchregister.py:123-128).
        '''
        if self.theta is None:
            raise RuntimeError("call optim() first")
        if self.mode == 'flow':
            return self.warp(moving)
        return self.warp(self.theta.detach(), moving)
