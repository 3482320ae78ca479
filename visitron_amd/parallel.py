"""Single-process data parallelism for the HIP modules: the reference's `multi-gpu-dp` launch mode
(tasks/viewpoint_select/pretrain.py:93-94, `model = torch.nn.DataParallel(model)`; the loop around it :169-193 then takes
`loss.mean()` of the per-GPU losses, calls `loss.backward()`, clips `model.parameters()` and steps the optimizer).

`torch.nn.DataParallel` re-creates the wrapped module on every forward as replicas whose weights are broadcast tensors; the
HIP modules keep per-device state between calls (the flat parameter slabs, packed bf16 weight copies, workspaces, saved
activations of the fused step), so they refuse such replicas (`modeling._refuse_data_parallel_replica`).  This class is the
drop-in for that one line: the same constructor, `.module`, `state_dict()` keys ("module." prefix), scatter along `dim`,
gather on `output_device`, gradients in the wrapped module's `.grad` after `backward()` -- built the way this package is
built:

* one PERSISTENT replica per extra device (constructed once from the module's class + config + state dict, living on its
  device with its own engine), refreshed from the master's parameters and buffers before every forward (one multi-tensor
  copy per replica: the bytes `torch.nn.DataParallel`'s broadcast moves as well);
* the replicas' forwards are enqueued one after the other from the calling thread -- each device's kernels run
  asynchronously, the host needs a few ms per replica against tens of ms of device time, and the library's process-wide
  state (autotuner, workspaces) is never entered from two threads;
* the outputs are gathered with torch's differentiable `Gather`, so `loss.backward()` runs each replica's backward on its
  own device into the replica's own parameters; a callback at the END of that backward adds the replicas' gradients into
  the master's `.grad` (on `device_ids[0]`) and clears them -- what `torch.nn.DataParallel`'s broadcast backward does.

`device_ids` may name one device several times (`[0, 0]`): a rehearsal of the whole mechanism on one GPU, which is how the
tests run it (tests/test_gpu_round6.py); on distinct devices the copies and the gather cross xGMI.  One process per GPU
(DistributedDataParallel / PretrainEngine with torch.distributed) remains the faster mode and the one
bench.py measures: this mode pays the weight refresh (440 MB per replica and step at the base config) and a serial gradient
reduction on device 0."""
import torch
from torch import nn

__all__ = ["DataParallel"]


def _scatter(obj, n, devices, dim):
    """obj -> list of n per-replica objects: tensors are chunked along `dim` (torch.chunk: the first chunks are the larger
    ones, there may be fewer than n) and moved to their device; tuples / lists / dicts are walked; anything else is
    handed to every replica as it is.  Returns (parts, count) with count = the number of replicas that got data."""
    if isinstance(obj, torch.Tensor):
        chunks = obj.chunk(n, dim) if obj.dim() > 0 else [obj] * n
        return [c.to(devices[i], non_blocking=True) for i, c in enumerate(chunks)], len(chunks)
    if isinstance(obj, (tuple, list)) and len(obj) > 0:
        parts = [_scatter(o, n, devices, dim) for o in obj]
        count = min([c for _, c in parts] or [n])
        return [type(obj)(p[i] for p, _ in parts) for i in range(count)], count
    if isinstance(obj, dict) and len(obj) > 0:
        parts = {k: _scatter(v, n, devices, dim) for k, v in obj.items()}
        count = min([c for _, c in parts.values()] or [n])
        return [type(obj)((k, p[i]) for k, (p, _) in parts.items()) for i in range(count)], count
    return [obj] * n, n


def _gather(outputs, device, dim):
    """Per-replica outputs -> one object on `device`: tensors through torch's differentiable Gather (0-d tensors become a
    vector with one entry per replica, as under torch.nn.DataParallel: the caller takes `.mean()`, pretrain.py:171-172),
    tuples / lists / dicts walked, None kept, Python numbers (the 7-tuple's `next_loss = 0` without next_action,
    encoder.py:395) as a vector of them."""
    from torch.nn.parallel._functions import Gather

    out = outputs[0]
    if isinstance(out, torch.Tensor):
        return Gather.apply(device, dim, *outputs)
    if out is None:
        return None
    if isinstance(out, dict):
        return type(out)((k, _gather([o[k] for o in outputs], device, dim)) for k in out)
    if isinstance(out, (tuple, list)):
        return type(out)(_gather(list(col), device, dim) for col in zip(*outputs))
    if isinstance(out, (int, float)):
        return torch.tensor([float(o) for o in outputs], device=device)
    return out


class DataParallel(nn.Module):
    """`torch.nn.DataParallel` for visitron_amd modules (see the module docstring).  `module` must be a model of this
    package that can be rebuilt from `type(module)(module.config)` + its state dict (PreTrainOscar,
    BertImgModelwithLocationEmbeds) and live on `device_ids[0]`."""

    def __init__(self, module, device_ids=None, output_device=None, dim=0):
        super().__init__()
        if not torch.cuda.is_available():
            raise RuntimeError("visitron_amd.parallel.DataParallel needs a GPU (the HIP modules have no CPU path)")
        if device_ids is None:
            device_ids = list(range(torch.cuda.device_count()))
        ids = [torch.device("cuda", d).index if isinstance(d, int) else torch.device(d).index for d in device_ids]
        if not ids:
            raise ValueError("device_ids is empty")
        if output_device is None:
            output_device = ids[0]
        self.module = module
        self.device_ids = ids
        self.output_device = torch.device("cuda", output_device if isinstance(output_device, int) else torch.device(output_device).index)
        self.dim = dim
        self.src_device = torch.device("cuda", ids[0])
        object.__setattr__(self, "_replicas", None)       # not sub-modules: parameters() / state_dict() show the master only
        object.__setattr__(self, "_reduce_pending", False)

    # ---- replicas -------------------------------------------------------------------------------------------------
    def _check_master(self):
        for t in list(self.module.parameters()) + list(self.module.buffers()):
            if t.device != self.src_device:
                raise RuntimeError("module must have its parameters and buffers on device %s (device_ids[0]) but found one "
                                   "of them on device: %s" % (self.src_device, t.device))

    def _build_replica(self, device):
        cfg = getattr(self.module, "config", None)
        if cfg is None:
            raise TypeError("visitron_amd.parallel.DataParallel wraps models built from a config (PreTrainOscar, "
                            "BertImgModelwithLocationEmbeds); got %s" % type(self.module).__name__)
        from .modeling import _is_fp32, set_precision

        with torch.cuda.device(device):
            r = type(self.module)(cfg)
            r.load_state_dict(self.module.state_dict())
            r.to(device)
        for (_, pm), (_, pr) in zip(self.module.named_parameters(), r.named_parameters()):
            pr.requires_grad_(pm.requires_grad)
        if _is_fp32(self.module):
            set_precision(r, "fp32")
        return r

    def replicas(self):
        """[master, replica on device_ids[1], ...] -- built on first use, kept."""
        if self._replicas is None:
            self._check_master()
            object.__setattr__(self, "_replicas",
                               [self.module] + [self._build_replica(torch.device("cuda", d)) for d in self.device_ids[1:]])
        return self._replicas

    def _refresh(self, reps):
        """Parameters, buffers and mode of the master into every other replica (one multi-tensor copy each)."""
        src_p = [p.data for p in self.module.parameters()]
        src_b = [b.data for b in self.module.buffers()]
        for r in reps[1:]:
            with torch.no_grad():
                torch._foreach_copy_([p.data for p in r.parameters()], src_p)
                if src_b:
                    torch._foreach_copy_([b.data for b in r.buffers()], src_b)
            if r.training != self.module.training:
                r.train(self.module.training)

    # ---- gradients ------------------------------------------------------------------------------------------------
    def _reduce_grads(self):
        """End of a backward: every replica's .grad added into the master's (on device_ids[0]) and dropped."""
        object.__setattr__(self, "_reduce_pending", False)
        reps = self._replicas or []
        with torch.no_grad():
            for r in reps[1:]:
                dst, src, fresh = [], [], []
                for pm, pr in zip(self.module.parameters(), r.parameters()):
                    if pr.grad is None:
                        continue
                    g = pr.grad.to(self.src_device, non_blocking=True)
                    if pm.grad is None:
                        fresh.append((pm, g))
                    else:
                        dst.append(pm.grad)
                        src.append(g)
                    pr.grad = None
                if dst:
                    torch._foreach_add_(dst, src)
                for pm, g in fresh:
                    pm.grad = g          # (a same-device replica's tensor itself: the replica has let go of it above)

    def _arm(self, tensor):
        """A hook on a gathered output that takes part in the backward: it queues ONE end-of-backward callback."""
        def hook(grad):
            if not self._reduce_pending:
                object.__setattr__(self, "_reduce_pending", True)
                torch.autograd.Variable._execution_engine.queue_callback(self._reduce_grads)
            return grad
        tensor.register_hook(hook)

    # ---- forward --------------------------------------------------------------------------------------------------
    def forward(self, *inputs, **kwargs):
        n = len(self.device_ids)
        if n == 1:
            return self.module(*inputs, **kwargs)
        reps = self.replicas()
        devices = [torch.device("cuda", d) for d in self.device_ids]
        ins, n_in = _scatter(tuple(inputs), n, devices, self.dim) if inputs else ([()] * n, n)
        kws, n_kw = _scatter(dict(kwargs), n, devices, self.dim) if kwargs else ([{}] * n, n)
        used = min(n_in, n_kw)
        self._refresh(reps[:used])
        outs = []
        for k in range(used):
            with torch.cuda.device(devices[k]):
                outs.append(reps[k](*ins[k], **kws[k]))
        out = _gather(outs, self.output_device, self.dim)
        if torch.is_grad_enabled() and used > 1:
            stack = [out]
            while stack:
                o = stack.pop()
                if isinstance(o, torch.Tensor):
                    if o.requires_grad:
                        self._arm(o)
                elif isinstance(o, dict):
                    stack.extend(o.values())
                elif isinstance(o, (tuple, list)):
                    stack.extend(o)
        return out
