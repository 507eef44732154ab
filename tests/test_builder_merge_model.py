"""CPU check of the Huffman builder's two-queue merge as build_dynamic_kernel runs it since round 5 (fpng_amd/csrc/kernels.hip,
dev_build_table): the reference takes its picks one after the other (src/fpng.cpp:645-651); the kernel's wave takes runs of picks at
once, and the length limiter behind it (:663-674) whole depth-first walks in one go.  tests/cpp/merge_model.cpp holds both forms of both
loops and compares parents, weights and the limited length counts (limits 7 and 12) on shapes of real histograms, on every multiset of
up to nine keys from a small set and on 150 000 random key sets (sums that wrap at 16 bits included); the kernel itself is held against the reference's files by the 2-pass GPU parity tests."""
import os
import subprocess

from cpu_ref import ROOT


def test_wave_parallel_merge_equals_the_serial_one(tmp_path):
    exe = str(tmp_path / "merge_model")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "merge_model.cpp")])
    for burst in ("1", "4"):
        out = subprocess.run([exe, burst], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        assert "bulk == serial" in out.stdout.splitlines()[-1]
