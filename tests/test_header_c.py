"""CPU-only: include/imgproc.h is valid C11 and valid C++11 on its own (what a node maintainer compiles against),
and a C translation unit that uses every type links against the built library."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include "imgproc.h"
#include <stdio.h>
int main(void) {
    imgproc_config cfg = {0, 64, 64, 2, 1};
    imgproc_pipeline p = {0};
    imgproc_result res;
    imgproc_ctx *ctx = 0;
    uint8_t elem[9];
    int32_t taps[5];
    p.n_morph = 1; p.morph[0].kind = IMGPROC_ERODE; p.morph[0].shape = IMGPROC_RECT; p.morph[0].kw = p.morph[0].kh = 3;
    (void)res; (void)cfg;
    if (imgproc_version() != IMGPROC_VERSION) return 1;
    if (imgproc_structuring_element(IMGPROC_RECT, 3, 3, elem) != IMGPROC_OK || elem[4] != 1) return 2;
    if (imgproc_gaussian_taps(5, taps) != IMGPROC_OK || taps[2] != 96) return 3;
    if (imgproc_set_pipeline(ctx, &p) != IMGPROC_EINVAL) return 4;   /* null context is rejected, not dereferenced */
    printf("%s\n", imgproc_strerror(IMGPROC_ENODEV));
    return 0;
}
'''


def test_header_compiles_as_c_and_cxx_and_links():
    inc = os.path.join(ROOT, "include")
    libdir = os.path.join(ROOT, "image_processing_amd", "lib")
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(SRC)
        subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-I", inc, c])
        subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Wextra", "-fsyntax-only", "-x", "c++", "-I", inc, c])
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-std=c11", "-I", inc, c, "-o", exe, "-L", libdir, "-limgproc_hip",
                               "-Wl,-rpath," + libdir])
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0, out
        assert "no usable HIP device" in out.stdout
