"""The boundary is a C ABI: include/ctdet.h must be plain C (no C++ or torch types) and a C program must link
against libctdet.so and get the known answers from the host entry points (no GPU needed)."""
import os
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(REPO, 'context-transformer_amd', 'lib')


@pytest.mark.skipif(shutil.which('gcc') is None, reason='gcc not available')
def test_header_is_c99_and_c_caller_links(tmp_path):
    assert os.path.exists(os.path.join(LIBDIR, 'libctdet.so')), 'build the library first (context-transformer_amd/build.py)'
    exe = str(tmp_path / 'c_caller')
    cmd = ['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-I' + os.path.join(REPO, 'include'),
           os.path.join(REPO, 'examples', 'c_caller.c'), '-L' + LIBDIR, '-lctdet', '-Wl,-rpath,' + LIBDIR, '-o', exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert 'cpu_nms keeps 2: 0 2' in r.stdout
