"""MODEL_REGISTRY / build_model -- the reference's model factory surface.

Same names, arguments and assertions as slowfast/models/build.py:8-55 so the reference's
train/test loops (tools/train_net.py:661, tools/test_net.py:203, scripts/module_wrapper.py:466-467)
can call this factory unchanged.
"""
import torch


class Registry(object):
    """Minimal name -> class registry with the fvcore interface the reference uses
    (``@REG.register()`` decorator and ``REG.get(name)``)."""

    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def _do_register(self, name, obj):
        assert name not in self._obj_map, "An object named '%s' was already registered in '%s' registry!" % (name, self._name)
        self._obj_map[name] = obj

    def register(self, obj=None):
        if obj is None:
            def deco(func_or_class):
                self._do_register(func_or_class.__name__, func_or_class)
                return func_or_class
            return deco
        self._do_register(obj.__name__, obj)

    def get(self, name):
        ret = self._obj_map.get(name)
        if ret is None:
            raise KeyError("No object named '%s' found in '%s' registry!" % (name, self._name))
        return ret

    def __contains__(self, name):
        return name in self._obj_map


MODEL_REGISTRY = Registry("MODEL")
MODEL_REGISTRY.__doc__ = "Registry for video models: the registered object is called as obj(cfg) and returns an nn.Module."


def build_model(cfg, gpu_id=None):
    """Builds the video model named by cfg.MODEL.MODEL_NAME (slowfast/models/build.py:17-55).

    ``torch.cuda`` is the ROCm device API on PyTorch-ROCm; DDP's "nccl" backend is RCCL.
    """
    if torch.cuda.is_available():
        assert cfg.NUM_GPUS <= torch.cuda.device_count(), "Cannot use more GPU devices than available"
    else:
        assert cfg.NUM_GPUS == 0, "Cuda is not available. Please set `NUM_GPUS: 0 for running on CPUs."
    model = MODEL_REGISTRY.get(cfg.MODEL.MODEL_NAME)(cfg)
    if cfg.NUM_GPUS:
        cur_device = torch.cuda.current_device() if gpu_id is None else gpu_id
        model = model.cuda(device=cur_device)
    if cfg.NUM_GPUS > 1:
        model = wrap_ddp(model, cfg, cur_device)
    return model


def wrap_ddp(model, cfg, device):
    """The reference's DistributedDataParallel wrap (build.py:47-54) with the knobs that matter on xGMI: gradients are views
    into the all-reduce buckets (no copy in / out of the 141 MB payload), the graph is declared static (all 350 parameters
    receive a gradient every step: no unused-parameter search, bucket order fixed after the first iteration), and optionally
    (HIP.DDP_BF16_GRADS) the payload is compressed to bf16 for the all-reduce and decompressed into the fp32 gradients."""
    hip = getattr(cfg, "HIP", None)
    model = torch.nn.parallel.DistributedDataParallel(
        module=model, device_ids=[device], output_device=device,
        gradient_as_bucket_view=bool(getattr(hip, "DDP_BUCKET_VIEW", True)),
        static_graph=bool(getattr(hip, "DDP_STATIC_GRAPH", True)))
    if bool(getattr(hip, "DDP_BF16_GRADS", False)):
        from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
        model.register_comm_hook(None, default_hooks.bf16_compress_hook)
    return model
