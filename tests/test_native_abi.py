"""CPU: the C-ABI library builds/loads, exports every symbol the header declares, validates
arguments before touching the device, and the product path refuses to run without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from openmatch_amd import native as N
from tests.conftest import REPO


def header_symbols():
    text = open(os.path.join(REPO, "include", "openmatch_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(om_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = N.lib()
    declared = header_symbols()
    assert len(declared) >= 12
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/openmatch_hip.h but not exported"
    assert sorted(N.exported_symbols()) == declared, "native.py binds a different symbol set than the header declares"
    assert lib.om_abi_version() == N.ABI_VERSION


def test_argument_validation_is_host_side():
    lib = N.lib()
    # K not a multiple of 128 bytes -> error before any launch
    rc = lib.om_gemm_nt(N.OM_F32, 16, 33, 16, 33, N.OM_F32, 16, 8, 4, 8, 33, None, None, 0, 0, None)
    assert rc != 0 and b"128" in lib.om_last_error()
    rc = lib.om_sim_topk(0, 16, 4, 16, None, None, 100, 768, 5000, 0, 16, 16, 256, 1 << 20, None)
    assert rc != 0 and b"2048" in lib.om_last_error()
    rc = lib.om_topk_merge(16, 16, 9, 4, 1000, 1000, 16, 16, None)
    assert rc != 0 and b"8192" in lib.om_last_error()
    assert lib.om_sim_topk_workspace_bytes(128, 768, 100) > 128 * 8192 * 8


def test_encoder_workspace_and_config_struct():
    lib = N.lib()
    cfg = N.OmEncoderConfig(arch=N.ARCH_BERT, dtype=N.OM_BF16, hidden=768, n_layers=12, n_heads=12, head_dim=64,
                            ffn=3072, vocab=30522, max_pos=512, type_vocab=2, act=N.ACT_GELU_ERF, ln_eps=1e-12)
    b1 = lib.om_encoder_workspace_bytes(C.byref(cfg), 256, 128)
    m = 256 * 128
    assert b1 >= m * (768 * 7 + 3072) * 2
    cfg.dtype = N.OM_F32
    assert lib.om_encoder_workspace_bytes(C.byref(cfg), 256, 128) > b1


def test_t5_bucket_host_function_matches_hf(golden):
    g = golden("t5_buckets")
    lib = N.lib()
    got = np.array([lib.om_t5_relative_bucket(int(r), 32, 128) for r in g["rel"]])
    assert (got == g["bucket"]).all()


def test_product_path_refuses_cpu_tensors():
    from openmatch_amd.modeling import DRModelForInference
    from tests.helpers import NS, tiny_bert_config
    from transformers import BertModel
    model = DRModelForInference(lm_q=(lm := BertModel(tiny_bert_config())), lm_p=lm, model_args=NS(encoder_only=False, dtype="float32"))
    batch = {"input_ids": torch.ones(2, 8, dtype=torch.long), "attention_mask": torch.ones(2, 8, dtype=torch.long)}
    with pytest.raises(N.NativeError, match="no CPU"):
        model(passage=batch)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(N, "_lib", None)
    monkeypatch.setattr(N, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(N.NativeError, match="missing"):
        N.lib()
