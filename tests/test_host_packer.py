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
    // whole records (ragged lengths, empty ones; one separator behind each): the record loop inside the AVX2 function against the generic one
    for (int iter = 0; iter < 2000; ++iter) {
        const size_t nrec = 1 + rand() % 30;
        std::vector<unsigned long long> off(nrec + 1, 0);
        for (size_t r = 0; r < nrec; ++r) off[r + 1] = off[r] + (rand() % 5 == 0 ? rand() % 4 : rand() % 200);
        std::vector<unsigned char> bytes(off[nrec] + 64);
        for (auto &c : bytes) c = (unsigned char)al[rand() % (sizeof(al) - 1)];
        const size_t ngmax = (off[nrec] + nrec + 15) / 16 + 2;
        std::vector<unsigned> c1(ngmax, 7), c2(ngmax, 9);
        std::vector<unsigned short> v1(ngmax, 7), v2(ngmax, 9);
        const size_t r0 = rand() % nrec, g1 = pack_records_generic(bytes.data(), off.data(), r0, nrec, c1.data(), v1.data());
        // the stream spelt out, encoded group by group with the scalar encoder
        std::vector<unsigned char> stream;
        for (size_t r = r0; r < nrec; ++r) { stream.insert(stream.end(), bytes.begin() + off[r], bytes.begin() + off[r + 1]); stream.push_back('\n'); }
        while (stream.size() % 16) stream.push_back('\n');
        if (g1 != stream.size() / 16) { printf("generic: %zu groups, want %zu\n", g1, stream.size() / 16); return 1; }
        for (size_t g = 0; g < g1; ++g) {
            unsigned c; unsigned short v;
            encode16_scalar(stream.data() + 16 * g, &c, &v);
            if (c != c1[g] || v != v1[g]) { printf("generic differs from the spelt-out stream: group %zu\n", g); return 1; }
        }
        if (__builtin_cpu_supports("avx2")) {
            const size_t g2 = pack_records_avx2(bytes.data(), off.data(), r0, nrec, c2.data(), v2.data());
            if (g2 != g1) { printf("avx2 records: %zu groups, want %zu\n", g2, g1); return 1; }
            for (size_t g = 0; g < g1; ++g) if (c1[g] != c2[g] || v1[g] != v2[g]) { printf("avx2 records differ: group %zu of %zu\n", g, g1); return 1; }
        }
    }
    if (getenv("PACK_TIMING")) {   // one thread's rate on 150-base records
        const size_t nrec = 400000;
        std::vector<unsigned long long> off(nrec + 1, 0);
        for (size_t r = 0; r < nrec; ++r) off[r + 1] = off[r] + 150;
        std::vector<unsigned char> bytes(off[nrec] + 64);
        for (auto &c : bytes) c = (unsigned char)"ACGT"[rand() & 3];
        std::vector<unsigned> c1(off[nrec] / 16 + nrec);
        std::vector<unsigned short> v1(c1.size());
        for (int which = 0; which < 2; ++which) {
            double best = 1e9;
            for (int rep = 0; rep < 7; ++rep) {
                const auto t0 = std::chrono::steady_clock::now();
                if (which) pack_records_avx2(bytes.data(), off.data(), 0, nrec, c1.data(), v1.data());
                else pack_records_generic(bytes.data(), off.data(), 0, nrec, c1.data(), v1.data());
                best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
            }
            printf("%s: %.2f GB/s of bases on one thread\n", which ? "record loop inside the AVX2 function" : "three calls per record", off[nrec] / best / 1e9);
        }
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
    cpp.write_text("#include <immintrin.h>\n#include <tmmintrin.h>\n#include <algorithm>\n#include <chrono>\n#include <cstdio>\n#include <cstdlib>\n#include <cstring>\n#include <vector>\n" + block + MAIN)
    exe = tmp_path / "enc"
    subprocess.run([CLANG, "-O3", "-std=c++17", "-o", str(exe), str(cpp)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr
    if os.environ.get("PACK_TIMING"):
        print(out.stdout)
