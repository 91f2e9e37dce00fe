"""The host packer's SIMD encoders (oxli_amd/csrc/kct_entry.hip: 16 / 32 ASCII bases -> 2-bit codes + validity bits, what kct_consume_batch
uploads instead of text) against the scalar encoder, on the CPU: the block between the [host-packer-begin] / [host-packer-end] markers is
compiled alone with the image's clang at -O3 -- the optimisation level the library is built with: one clang miscompiled two
__builtin_bitreverse16 of a 32-bit mask's halves there (the halves came out swapped), which only messy input shows."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"

MAIN = r"""
int main() {
    srand(1);
    const char al[] = "ACGTacgtNn-\xc3\xa9R";
    for (int iter = 0; iter < 4000; ++iter) {
        size_t ng = 1 + rand() % 40;
        std::vector<unsigned char> buf(16 * ng + 32);
        for (auto &c : buf) c = (unsigned char)al[rand() % (sizeof(al) - 1)];
        std::vector<unsigned> c1(ng), c2(ng);
        std::vector<unsigned short> v1(ng), v2(ng);
        for (size_t g = 0; g < ng; ++g) encode16_scalar(buf.data() + 16 * g, &c1[g], &v1[g]);
        if (__builtin_cpu_supports("avx2")) {
            encode_run_avx2(buf.data(), ng, c2.data(), v2.data());
            for (size_t g = 0; g < ng; ++g) if (c1[g] != c2[g] || v1[g] != v2[g]) { printf("avx2 differs: group %zu of %zu\n", g, ng); return 1; }
        }
        if (__builtin_cpu_supports("ssse3")) {
            encode_run_ssse3(buf.data(), ng, c2.data(), v2.data());
            for (size_t g = 0; g < ng; ++g) if (c1[g] != c2[g] || v1[g] != v2[g]) { printf("ssse3 differs: group %zu of %zu\n", g, ng); return 1; }
        }
        encode_groups(buf.data(), ng, c2.data(), v2.data());
        for (size_t g = 0; g < ng; ++g) if (c1[g] != c2[g] || v1[g] != v2[g]) { printf("dispatch differs\n"); return 1; }
    }
    puts("ok");
    return 0;
}
"""


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm clang")
def test_simd_encoders_equal_the_scalar_encoder(tmp_path):
    src = open(os.path.join(ROOT, "oxli_amd", "csrc", "kct_entry.hip")).read()
    block = src[src.index("// [host-packer-begin]"):src.index("// [host-packer-end]")]
    cpp = tmp_path / "enc.cpp"
    cpp.write_text("#include <immintrin.h>\n#include <tmmintrin.h>\n#include <cstdio>\n#include <cstdlib>\n#include <cstring>\n#include <vector>\n" + block + MAIN)
    exe = tmp_path / "enc"
    subprocess.run([CLANG, "-O3", "-std=c++17", "-o", str(exe), str(cpp)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr
