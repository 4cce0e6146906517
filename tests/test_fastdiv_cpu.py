"""The divide-free exact quotient of the single-launch kernels (csrc/cnnq_qdq.hip.h `qdq1_fast`: the channel's correctly
rounded reciprocal and two fma corrections) against the C divide on the host: tests/c/fastdiv_check.c restates the
sequence with fmaf and brute-forces ~1e8 (dividend, scale) pairs - random ones, dividends a few ulps around the rounding
ties of the quotient and of the integer code, all-ones and power-of-two significands, the scale floor, zeros and
denormals.  Inside the domain the kernels check, quotients must be bit-identical from 2^-70 up and codes / dequantized
values identical below.  (The -m gpu parity tests then compare the kernels' outputs with the oracle bit for bit.)"""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(shutil.which('gcc') is None, reason='gcc not found')
def test_divide_free_quotient_is_the_ieee_quotient(tmp_path):
    exe = str(tmp_path / 'fastdiv_check')
    flags = ['-O2', '-ffp-contract=off']
    try:
        if ' fma ' in open('/proc/cpuinfo').read():
            flags.append('-mfma')            # fmaf as one instruction; without it glibc's exact software fmaf (slower)
    except OSError:
        pass
    subprocess.run(['gcc'] + flags + [os.path.join(HERE, 'c', 'fastdiv_check.c'), '-o', exe, '-lm'], check=True)
    n = (4000, 25000) if '-mfma' in flags else (400, 5000)
    r = subprocess.run([exe, str(n[0]), str(n[1]), '7'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert ' mismatches 0 ' in r.stdout and r.stdout.rstrip().endswith('mismatches 0'), r.stdout
