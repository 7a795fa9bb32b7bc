"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports every symbol
that include/xmlhip.h declares (no compute calls: there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from tvretrieval_amd import _lib
    if not os.path.isfile(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


def header_symbols():
    src = open(os.path.join(ROOT, "include", "xmlhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(xml_[a-z0-9_]+)\s*\(", src)))


def test_every_header_symbol_is_exported_and_bound(lib):
    from tvretrieval_amd import _lib
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "libxmlhip.so does not export %s" % s
        assert s in _lib.SIGNATURES, "ctypes binding missing for %s" % s
    assert set(_lib.SIGNATURES) <= set(syms), set(_lib.SIGNATURES) - set(syms)


def test_abi_identity(lib):
    from tvretrieval_amd import _lib
    assert lib.xml_abi_version() == _lib.ABI_VERSION == 6
    assert lib.xml_build_arch() == b"gfx950"
    assert lib.xml_status_string(0) == b"ok"
    assert lib.xml_status_string(-2) == b"unsupported shape"


def test_argument_validation_without_gpu(lib):
    """Entry points validate before launching: bad arguments return error codes, never crash."""
    import ctypes
    from tvretrieval_amd._lib import ConvseDesc
    assert lib.xml_q2c_scores(None, None, None, None, 0, 1, 1, 16, 8, 0, 0, None) == -1
    assert lib.xml_topk_rows(None, 0, None, None, None, 1, 1, 1, 0.0, None, 0, None) == -1
    assert lib.xml_linear(None, None, None, None, 1, 8, 8, 0, 0, None) == -1
    d = ConvseDesc(nq=10, nv=20, kpairs=5, lpad=128, l_ref=100, hidden=768, n_mod=2, merged=1, ksize=5, softmax=1, dt=1)
    assert lib.xml_convse_rerank_workspace_bytes(ctypes.byref(d)) > 20 * 4
    assert lib.xml_attention_block_workspace_bytes(4, 128, 768, 1) >= 4 * 128 * 768 * (3 * 2 + 2 + 4)
    assert lib.xml_linear_ln_relu_pos_workspace_bytes(512, 3072, 768, 0) >= 512 * (3072 + 768) * 4


def test_round3_entries_validate_arguments(lib):
    """The exact-rank, packed-sequence and accumulate entries added in round 3 (ABI 3) reject bad arguments before any launch."""
    assert lib.xml_round_bf16_rows_err(None, None, None, 4, 768, None) == -1
    assert lib.xml_q2c_rescore_workspace_bytes(10000, 21793, 256) >= (10000 * 256 + 2 * 21793) * 4
    assert lib.xml_q2c_rescore(2, None, None, None, None, None, None, None, None, 1, 1, 1, 128, 768, 0, None, 0, None) == -1
    assert lib.xml_exact_certificate(None, 256, None, 100, None, None, 0.0, 0.0, 2, 0.0, 20.0, 1, None, None, None, None, 10,
                                     None) == -1
    assert lib.xml_select_ge_rows(None, 0, None, None, 0, None, 1, 1, None) == -1
    assert lib.xml_attention_block_varlen_workspace_bytes(175000, 768, 1) >= 175000 * 768 * (3 * 2 + 2)
    assert lib.xml_attention_block_varlen(None, None, None, None, None, None, None, None, None, 10, 2, 30, 768, 4, 1, None, 0,
                                          None) == -1
    assert lib.xml_modular_pool_varlen(None, None, None, None, 2, 30, 768, 2, 1, None) == -1
    assert lib.xml_gemm_tn(None, None, None, None, 100, 768, 768, 1, 1, None) == -1
    assert lib.xml_pack_plan(None, 10, 30, None, None, None, None) == -1
    assert lib.xml_linear_ln_relu_pos_packed_workspace_bytes(175000, 768, 768, 1) > \
        lib.xml_linear_ln_relu_pos_workspace_bytes(175000, 768, 768, 1)
    assert lib.xml_linear_ln_relu_pos_packed(None, 0, None, 30, None, None, None, None, None, None, None, None, 10, 768, 768,
                                             1, None, 0, None) == -1


def test_round5_entries_validate_arguments(lib):
    """K10 (xml_moments_decode) and the batched host NMS (ABI 5) reject bad arguments before any launch / thread."""
    assert lib.xml_moments_decode(None, None, None, None, None, 4, 8, 8, 2, 16, 1.5, 1, None, 8, None, None) == -1
    assert lib.xml_nms_vcmr_batched_host(None, None, None, None, None, 1, 8, 0.5, 100, 100, None, 100, None, 0) == -1
    assert lib.xml_nms_svmr_batched_host(None, None, None, None, 1, 8, 0.5, 100, 100, None, 100, None, 0) == -1


def test_product_path_fails_loudly_without_gpu():
    """No CPU fallback: CPU tensors are rejected by the ops layer."""
    import torch
    from tvretrieval_amd import _lib, ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.XmlHipError):
        ops.l2norm_rows(torch.zeros(4, 8))
    with pytest.raises(_lib.XmlHipError):
        ops.topk_rows(torch.zeros(4, 8), 2)


def test_checkpoint_layout_roundtrip(tmp_path):
    """{"model", "model_cfg", "epoch"} (xml/train.py:219-223) incl. the EasyDict config survives torch.save/load."""
    import torch
    from conftest import load_golden
    from tvretrieval_amd.easydict_compat import register_easydict_module
    from tvretrieval_amd.model_xml import XML
    register_easydict_module()
    d, cfg, sd = load_golden("xml_video_sub_cross_h128")
    m = XML(cfg)
    assert set(m.state_dict().keys()) == set(sd.keys())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    path = str(tmp_path / "model.ckpt")
    torch.save({"model": m.state_dict(), "model_cfg": m.config, "epoch": 3}, path)
    ck = torch.load(path, weights_only=False)
    assert ck["epoch"] == 3 and ck["model_cfg"].hidden_size == cfg["hidden_size"]
    ck["model_cfg"]["stack_conv_predictor_conv_kernel_sizes"] = -1      # xml/inference.py:538
    m2 = XML(ck["model_cfg"])
    m2.load_state_dict(ck["model"])
    for k, v in m2.state_dict().items():
        assert torch.equal(v, torch.from_numpy(sd[k]))


def test_config_is_validated_at_construction():
    """Shapes outside what the kernels implement fail in XML.__init__ with the reason (not deep inside a launch)."""
    import pytest
    from tvretrieval_amd.model_xml import XML, xml_base_config
    with pytest.raises(ValueError, match="multiple"):
        XML(dict(xml_base_config))                      # hidden_size = 500 placeholder of xml_base_config
    ok = dict(xml_base_config, hidden_size=256)
    XML(ok)
    with pytest.raises(ValueError, match="max_ctx_l"):
        XML(dict(ok, max_ctx_l=200))
    with pytest.raises(ValueError, match="conv_kernel_size"):
        XML(dict(ok, conv_kernel_size=4))


def test_config_pickles_as_easydict_module():
    """model.config inside a checkpoint must resolve as `easydict.EasyDict` on the reference side (xml/train.py:219-223)."""
    import pickle
    import pickletools
    from tvretrieval_amd.model_xml import XML, xml_base_config
    m = XML(dict(xml_base_config, hidden_size=256))
    blob = pickle.dumps(m.config, protocol=2)
    names = [arg for op, arg, _ in pickletools.genops(blob) if op.name == "GLOBAL"]
    assert names and all(n.startswith("easydict ") for n in names), names
    back = pickle.loads(blob)
    assert back.hidden_size == 256 and dict(back) == dict(m.config)


def test_split_f16_entries_validate_without_gpu(lib):
    """The split-f16 entries (ABI 4: exact-rank mode on the 16-bit pipe) reject bad arguments before any launch."""
    import ctypes
    p = ctypes.c_void_p(0x1000)
    z = ctypes.c_void_p(0)
    assert lib.xml_split_f16_rows(z, p, p, z, z, 4, 64, -1, z) == -1            # null input
    assert lib.xml_split_f16_rows(p, p, p, z, z, 4, 48, -1, z) == -2            # k % 32
    assert lib.xml_split_f16_rows(p, p, p, z, z, 4, 64, 99, z) == -1            # scale out of range
    assert lib.xml_unsplit_f16_rows(p, z, p, 4, 64, z) == -1
    assert lib.xml_pack_weights_f16s_bytes(8, 16) == 8 * 16 * 6 + 16 and lib.xml_pack_weights_f16s_bytes(0, 16) == 0
    assert lib.xml_pack_weights_f16s(p, p, 8, 12, z) == -2                      # k % 8
    assert lib.xml_linear_f16s(p, p, z, p, 4, 8, 16, 0, z, 0, z) == -1          # no workspace
    assert lib.xml_linear_f16s_workspace_bytes(100, 64) >= 100 * 64 * 6 + 400
    assert lib.xml_linear_f16s(p, p, z, p, 100, 8, 64, 0, p, 16, z) == -3       # workspace too small
    from tvretrieval_amd._lib import ConvseDesc, XML_F16S, XML_F32
    d = ConvseDesc(nq=2, nv=2, kpairs=1, lpad=16, l_ref=16, hidden=64, n_mod=1, merged=0, ksize=5, softmax=1, dt=XML_F32)
    assert lib.xml_convse_rerank_f16s(ctypes.byref(d), p, z, p, z, p, z, p, z, p, z, p, p, p, p, p, 1 << 20, z) == -1   # dt
    d.dt, d.hidden = XML_F16S, 40
    assert lib.xml_convse_rerank_f16s(ctypes.byref(d), p, z, p, z, p, z, p, z, p, z, p, p, p, p, p, 1 << 20, z) == -2   # hidden % 32
    # the model entries accept XML_F16S as a compute dtype and still validate their workspace
    assert lib.xml_attention_block_workspace_bytes(4, 16, 64, XML_F16S) > lib.xml_attention_block_workspace_bytes(4, 16, 64, XML_F32)
    assert lib.xml_q2c_tiled_ok(128, 768, 2) == 1 and lib.xml_q2c_tiled_ok(128, 768, 3) == 0
