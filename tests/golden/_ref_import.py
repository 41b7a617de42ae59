"""Imports the reference's modules in the BUILD CONTAINER ONLY (golden generation).

/root/reference cannot travel to the GPU box, so nothing under tests/ imports this at test time;
only tests/golden/make_goldens.py does.  `import diffsynth` as a package fails here (imageio,
modelscope, cv2, ... absent), so the packages are registered as empty stubs whose __path__ points at
the real directories and only the needed leaf modules are imported.
"""
import importlib
import importlib.machinery
import os
import sys
import types

sys.dont_write_bytecode = True  # never write __pycache__ into the read-only reference tree
REF = os.environ.get("GF_REFERENCE", "/root/reference")


def _stub(name, path=None, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=path is not None)
    if path is not None:
        m.__path__ = [path]
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Anything:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, n):
        return _Anything()


def load_reference(with_pipeline=True, with_dataset=True):
    if not os.path.isdir(REF):
        raise RuntimeError(f"reference tree not found at {REF}")
    ds = os.path.join(REF, "diffsynth")
    _stub("diffsynth", ds)
    _stub("diffsynth.models", os.path.join(ds, "models"))
    _stub("diffsynth.schedulers", os.path.join(ds, "schedulers"))
    _stub("diffsynth.vram_management", os.path.join(ds, "vram_management"))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    out = types.SimpleNamespace()
    out.dit = importlib.import_module("diffsynth.models.wan_video_dit")
    out.vae = importlib.import_module("diffsynth.models.wan_video_vae")
    out.fm = importlib.import_module("diffsynth.schedulers.flow_match")
    out.vram = importlib.import_module("diffsynth.vram_management.layers")
    out.t5 = importlib.import_module("diffsynth.models.wan_video_text_encoder")
    if with_pipeline:
        # wan_video_new.py needs names at import time only; all uses are at call time
        _stub("modelscope", snapshot_download=_Anything())
        _stub("diffsynth.utils", BasePipeline=object, ModelConfig=_Anything, PipelineUnit=object,
              PipelineUnitRunner=_Anything)
        sys.modules["diffsynth.models"].ModelManager = _Anything
        sys.modules["diffsynth.models"].load_state_dict = _Anything()
        _stub("diffsynth.models.wan_video_dit_s2v", rope_precompute=_Anything())
        _stub("diffsynth.models.wan_video_text_encoder", WanTextEncoder=_Anything, T5RelativeEmbedding=_Anything,
              T5LayerNorm=_Anything)
        _stub("diffsynth.models.wan_video_image_encoder", WanImageEncoder=_Anything)
        _stub("diffsynth.models.wan_video_vace", VaceWanModel=_Anything)
        _stub("diffsynth.models.wan_video_motion_controller", WanMotionControllerModel=_Anything)
        _stub("diffsynth.prompters", WanPrompter=_Anything)
        _stub("diffsynth.lora", GeneralLoRALoader=_Anything)
        vm = sys.modules["diffsynth.vram_management"]
        vm.enable_vram_management = _Anything()
        vm.AutoWrappedModule = out.vram.AutoWrappedModule
        vm.AutoWrappedLinear = out.vram.AutoWrappedLinear
        vm.WanAutoCastLayerNorm = out.vram.WanAutoCastLayerNorm
        sys.modules["diffsynth.schedulers"].flow_match = out.fm
        spec = importlib.util.spec_from_file_location("gf_ref_wan_video_new",
                                                      os.path.join(REF, "src/goal_force/wan_video_new.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        out.gf = mod
    if with_dataset:
        out.real = {}
        for n in ("torchvision", "imageio", "controlnet_aux", "cv2"):
            if n not in sys.modules:
                try:                                   # the real package when the image has it (g14 needs the real cv2 + controlnet_aux)
                    importlib.import_module(n)
                    out.real[n] = True
                except Exception:                      # noqa: BLE001 — absent (or broken) here: an empty stub, uses are at call time
                    _stub(n)
                    out.real[n] = False
        if not out.real.get("torchvision", True):
            tv = sys.modules["torchvision"]
            tv.transforms = _stub("torchvision.transforms", ToTensor=_Anything, ToPILImage=_Anything, Compose=_Anything,
                                  Resize=_Anything, CenterCrop=_Anything, Normalize=_Anything)
        if not out.real.get("imageio", True):
            sys.modules["imageio"].v3 = _stub("imageio.v3")
        if not out.real.get("controlnet_aux", True):
            sys.modules["controlnet_aux"].CannyDetector = _Anything
        spec = importlib.util.spec_from_file_location("gf_ref_unified_dataset",
                                                      os.path.join(REF, "src/goal_force/unified_dataset.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        out.ds = mod
    return out
