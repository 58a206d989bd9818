"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/comap_mi355x.h
declares, validates its arguments like the reference (exceptions -> status codes), and FAILS LOUDLY without a GPU
(no CPU fallback exists)."""
import ctypes
import os
import re

import numpy as np
import pytest

from comap_amd import engine, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    txt = open(os.path.join(ROOT, "include", "comap_mi355x.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cmx_[a-z_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = engine.load_library()
    names = _header_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/comap_mi355x.h but not exported"
    assert sorted(engine.EXPORTS) == names
    assert b"gfx950" in lib.cmx_version()


def _tiny():
    parent, blen, lot = synthetic.random_tree(6, 3)
    m = synthetic.dna_model()
    return parent, blen, lot, m


def test_model_validation_errors_come_before_any_device_work():
    parent, blen, lot, m = _tiny()
    with pytest.raises(engine.CmxError) as e:
        engine.Engine(parent, blen, lot, np.eye(65) - 1, np.full(65, 1 / 65), m["rates"], m["probs"])
    assert e.value.status == -2 and "nstates" in str(e.value)            # CMX_ERR_UNSUPPORTED: more than 64 states
    bad_parent = parent.copy()
    bad_parent[0], bad_parent[1] = 0, 0
    with pytest.raises(engine.CmxError) as e:
        engine.Engine(bad_parent, blen, lot, m["Q"], m["pi"], m["rates"], m["probs"])
    assert e.value.status == -1 and "post-order" in str(e.value)
    Qbad = m["Q"].copy()
    Qbad[0, 1] *= 2
    with pytest.raises(engine.CmxError) as e:
        engine.Engine(parent, blen, lot, Qbad, m["pi"], m["rates"], m["probs"])
    assert e.value.status == -1


def test_no_gpu_means_loud_failure_not_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    parent, blen, lot, m = _tiny()
    with pytest.raises(engine.CmxError) as e:
        engine.Engine(parent, blen, lot, m["Q"], m["pi"], m["rates"], m["probs"])
    assert e.value.status == -3 and "no CPU path" in str(e.value)         # CMX_ERR_DEVICE


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "comap_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "liboracle" not in txt and not re.search(r"#include\s*[<\"].*oracle", txt), f
