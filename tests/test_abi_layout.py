"""CPU checks of the drop-in boundary: the shared library loads without a GPU, exports every
symbol that include/*.h declares, and the ctypes mirrors in 21cmfast_amd/structs.py agree
byte for byte with what the C compiler lays out for the headers."""

import ctypes as C
import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADERS = [ROOT / "include" / "c21cm_abi.h", ROOT / "include" / "c21cm_grid.h"]


def declared_functions():
    names = set()
    for h in HEADERS:
        text = re.sub(r"/\*.*?\*/", "", h.read_text(), flags=re.S)
        text = re.sub(r"typedef[^;]*\(\*[^;]*;", "", text)  # function-pointer typedefs
        for m in re.finditer(r"^[A-Za-z_][\w \*]*?\b(\w+)\s*\([^;{]*\)\s*;", text, flags=re.M):
            names.add(m.group(1))
    return sorted(names)


def test_library_loads_without_gpu_and_exports_every_declared_symbol(pkg):
    lib = pkg.load()
    names = declared_functions()
    assert {"ComputeInitialConditions", "ComputePerturbedField", "ComputeIonizedBox",
            "Broadcast_struct_global_all", "test_filter", "init_ps", "c21cm_ionize_grids",
            "c21cm_perturb_grids", "c21cm_ics_grids", "c21cm_ionize_shard_radii",
            "ComputeBrightnessTemp", "ComputeHaloBox", "test_halo_props", "UpdateXraySourceBox", "hyper_2F3",
            "c21cm_fill_Rbox_grids", "c21cm_annular_filter_grids"} <= set(names)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"
    for g in ("simulation_options_global", "matter_options_global", "cosmo_params_global",
              "astro_params_global", "astro_options_global", "cosmo_tables_global",
              "config_settings"):
        C.c_void_p.in_dll(lib, g)
    assert lib.c21cm_version().startswith(b"21cmfast_amd")


STRUCTS = ["CosmoParams", "SimulationOptions", "MatterOptions", "AstroParams", "AstroOptions",
           "CosmoTables", "ConfigSettings", "InitialConditions", "PerturbedField", "HaloBox",
           "TsBox", "IonizedBox", "c21cm_ionize_spec", "c21cm_ionize_report", "c21cm_perturb_spec",
           "c21cm_ics_spec", "BrightnessTemp", "XraySourceBox", "c21cm_brightness_spec",
           "c21cm_halobox_spec", "c21cm_rbox_spec", "c21cm_annular_spec", "HaloCatalog",
           "c21cm_halo_consts", "c21cm_mturn_spec", "c21cm_ts_spec", "c21cm_ts_report",
           "c21cm_ts_first_spec"]
PY_NAMES = {"InitialConditions": "InitialConditionsStruct", "PerturbedField": "PerturbedFieldStruct",
            "HaloBox": "HaloBoxStruct", "TsBox": "TsBoxStruct", "IonizedBox": "IonizedBoxStruct",
            "c21cm_ionize_spec": "IonizeSpec", "c21cm_ionize_report": "IonizeReport",
            "c21cm_perturb_spec": "PerturbSpec", "c21cm_ics_spec": "IcsSpec",
            "BrightnessTemp": "BrightnessTempStruct", "XraySourceBox": "XraySourceBoxStruct",
            "c21cm_brightness_spec": "BrightnessSpec", "c21cm_halobox_spec": "HaloBoxSpec",
            "c21cm_rbox_spec": "RboxSpec", "c21cm_annular_spec": "AnnularSpec",
            "HaloCatalog": "HaloCatalogStruct", "c21cm_halo_consts": "HaloConsts",
            "c21cm_mturn_spec": "MturnSpec", "c21cm_ts_spec": "TsSpec", "c21cm_ts_report": "TsReport",
            "c21cm_ts_first_spec": "TsFirstSpec"}


def test_ctypes_mirrors_match_compiler_layout(pkg, tmp_path):
    S = pkg.structs
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "c21cm_grid.h"', "int main(void){"]
    for name in STRUCTS:
        cls = getattr(S, PY_NAMES.get(name, name))
        lines.append(f'printf("{name} size %zu\\n", sizeof({name}));')
        for field, _ in cls._fields_:
            lines.append(f'printf("{name} {field} %zu\\n", offsetof({name}, {field}));')
    lines.append("return 0;}")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", str(ROOT / "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    for line in out.strip().splitlines():
        name, field, value = line.split()
        cls = getattr(S, PY_NAMES.get(name, name))
        if field == "size":
            assert C.sizeof(cls) == int(value), name
        else:
            assert getattr(cls, field).offset == int(value), f"{name}.{field}"


def test_boundary_matches_the_reference_headers_layout(tmp_path):
    """(b) pinned to the REFERENCE, not to this repo: tests/golden/abi_layout.json holds
    sizeof and every field's offset / size as gcc lays out the reference's own cffi headers
    (_inputparams_wrapper.h:6-202, _outputstructs_wrapper.h:6-105; written in the build container by
    tests/golden/make_abi_layout.py).  include/c21cm_abi.h compiled here must give the same numbers for every
    struct of the path -- a field added, dropped, reordered or retyped on either side fails this test.  Exported
    entry points the reference's prototypes name must keep their argument counts."""
    import json
    import sys

    sys.path.insert(0, str(ROOT / "tests" / "golden"))
    import make_abi_layout as M

    doc = json.loads((ROOT / "tests" / "golden" / "abi_layout.json").read_text())
    ref = doc["structs"]
    # outside SURVEY section 8 (the discrete halo sampler's catalogue of perturbed halos): not in the drop-in header
    out_of_scope = {"PerturbedHaloCatalog"}
    ours_text = (ROOT / "include" / "c21cm_abi.h").read_text()
    ours = dict(M.parse_structs(ours_text))
    wanted = [(name, [f[0] for f in rec["fields"]]) for name, rec in sorted(ref.items()) if name not in out_of_scope]
    missing = [name for name, _ in wanted if name not in ours]
    assert not missing, f"structs of the reference boundary absent from include/c21cm_abi.h: {missing}"
    for name, fields in wanted:  # same field NAMES in the same order
        assert ours[name] == fields, f"{name}: fields {ours[name]} != reference {fields}"
    got = M.layout_of(["c21cm_abi.h"], wanted, include_dirs=[str(ROOT / "include")])
    for name, _ in wanted:
        assert got[name]["size"] == ref[name]["size"], f"sizeof({name})"
        assert got[name]["fields"] == ref[name]["fields"], f"{name}: offsets / sizes differ"
    # prototypes: every reference-named function the header declares has the reference's argument count
    ours_protos = M.parse_prototypes(ours_text)
    shared = sorted(set(ours_protos) & set(doc["prototype_arg_counts"]))
    assert {"ComputeInitialConditions", "ComputePerturbedField", "ComputeIonizedBox", "ComputeTsBox",
            "ComputeBrightnessTemp", "ComputeHaloBox", "UpdateXraySourceBox",
            "Broadcast_struct_global_all"} <= set(shared)
    for fn in shared:
        assert ours_protos[fn] == doc["prototype_arg_counts"][fn], fn


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under 21cmfast_amd/ may reference it."""
    for path in (ROOT / "21cmfast_amd").rglob("*"):
        if path.suffix in {".py", ".c", ".h", ".hip"} or path.name == "Makefile":
            text = path.read_text(errors="ignore")
            assert "liboracle" not in text and "oracle_" not in text, path
            assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), path


def test_missing_extension_fails_loudly(pkg, monkeypatch, tmp_path):
    lib_mod = __import__("importlib").import_module("21cmfast_amd._lib")
    monkeypatch.setattr(lib_mod, "_lib", None)
    monkeypatch.setattr(lib_mod, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        lib_mod.load()
