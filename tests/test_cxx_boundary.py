"""Row (b), the drop-in boundary, as far as it can be checked without MFEM / a GPU:
  * include/saamge_amd.hpp + saamge_amd_mfem.hpp compile with SAAMGE_AMD_WITH_MFEM against the
    declaration-only stand-in tests/mfem_stub/mfem.hpp, through a mock driver that follows the call
    sequence of amg/test/mltest/mltest.cpp:667-793 (11-argument MultilevelParameters, ml_produce_data with
    an ElementMatrixProvider, levels_list_get_level, VCycleSolver incl. iterative_mode, assignable
    coarse_solver, kalchev_pcg, smpr_ft plug);
  * a C++ program links libsaamge_amd.so through saamge_amd::api and exercises the argument checks;
  * the C header compiles as C."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
HIP_INC = "/opt/rocm/include"


def _run(cmd, **kw):
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, **kw)
    assert p.returncode == 0, " ".join(cmd) + "\n" + p.stdout
    return p.stdout


def test_mfem_adaptor_header_compiles_against_stub(tmp_path):
    _run(["g++", "-std=c++14", "-Wall", "-Wextra", "-Werror", "-DSAAMGE_AMD_WITH_MFEM", "-I", INC,
          "-I", os.path.join(ROOT, "tests", "mfem_stub"), "-c", os.path.join(ROOT, "tests", "cxx", "mock_driver.cpp"),
          "-o", str(tmp_path / "mock_driver.o")])


def test_api_header_links_the_library(tmp_path):
    lib_dir = os.path.join(ROOT, "saamge_amd")
    assert os.path.exists(os.path.join(lib_dir, "libsaamge_amd.so")), "run __graft_entry__.build() first"
    exe = str(tmp_path / "api_link_test")
    _run(["g++", "-std=c++14", "-Wall", "-Wextra", "-Werror", "-I", INC, os.path.join(ROOT, "tests", "cxx", "api_link_test.cpp"),
          "-o", exe, "-L", lib_dir, "-lsaamge_amd", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    out = _run([exe])
    assert "api link test ok" in out


def test_c_header_is_plain_c(tmp_path):
    src = tmp_path / "c_abi.c"
    src.write_text('#include "saamge_amd.h"\nint main(void) { saamge_amd_params p; saamge_amd_params_default(&p); return p.num_coarsenings != 1; }\n')
    lib_dir = os.path.join(ROOT, "saamge_amd")
    exe = str(tmp_path / "c_abi")
    _run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", INC, str(src), "-o", exe, "-L", lib_dir, "-lsaamge_amd",
          "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    _run([exe])
