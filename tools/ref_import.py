"""Import the reference (jayleicn/TVRetrieval, mounted read-only at /root/reference) on CPU.

Development-container only: the reference never travels to the GPU box. This module is used by
tools/make_golden.py (fixture generation) and by the optional `ref`-marked tests that
cross-check the oracle against the live reference when /root/reference exists.

The reference needs two packages that are not installed here (`easydict`, `h5py`); only an
attribute-dict class and an `h5py.File` name used in isinstance() checks are touched, so tiny
in-memory module objects are registered in sys.modules.  numpy >= 1.24
removed `np.int` / `np.bool`, which xml/inference.py:289 and standalone_eval/eval.py still use.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("TVR_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "baselines", "crossmodal_moment_localization"))


def _install_stand_ins():
    import numpy as np
    if not hasattr(np, "int"):
        np.int = int
    if not hasattr(np, "bool"):
        np.bool = bool
    if "easydict" not in sys.modules:
        from tvretrieval_amd.easydict_compat import EasyDict
        mod = types.ModuleType("easydict")
        mod.EasyDict = EasyDict
        sys.modules["easydict"] = mod
    if "h5py" not in sys.modules:
        mod = types.ModuleType("h5py")

        class File(object):  # only used as isinstance(x, h5py.File)
            pass
        mod.File = File
        sys.modules["h5py"] = mod


def import_reference():
    """Returns a namespace with the reference modules used for fixture generation."""
    if not reference_available():
        raise RuntimeError("reference tree not found at %s" % REF_ROOT)
    sys.dont_write_bytecode = True
    _install_stand_ins()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    ns = types.SimpleNamespace()
    from baselines.crossmodal_moment_localization import model_xml, model_components, optimization
    ns.model_xml = model_xml
    ns.model_components = model_components
    ns.optimization = optimization
    from baselines.crossmodal_moment_localization import inference as xml_inference
    ns.inference = xml_inference
    from utils import temporal_nms, tensor_utils, basic_utils
    ns.temporal_nms = temporal_nms
    ns.tensor_utils = tensor_utils
    ns.basic_utils = basic_utils
    from baselines.clip_alignment_with_language import inference as cal_inference
    ns.cal_inference = cal_inference
    from standalone_eval import eval as standalone_eval
    ns.standalone_eval = standalone_eval
    return ns
