"""CPU check of the restatement build_dynamic_kernel (fpng_amd/csrc/kernels.hip) uses for the run-length packing of the Deflate
code lengths: the reference walks the sequence with a small state machine (pending zero run / pending repeat count, reference
src/fpng.cpp:711-726, :770-794); the kernel handles every MAXIMAL RUN of equal lengths on its own lane with a closed form.  Both
are written down here and compared on random sequences (tokens and the code-length histogram); the kernel itself is held against
the reference's files by the 2-pass GPU parity tests."""
import random


def state_machine(seq):
    packed, c2 = [], [0] * 19
    zrun = rep = 0
    prev = 0xFF

    def flush_rep():
        nonlocal rep
        if rep:
            if rep < 3:
                c2[prev] += rep
                packed.extend([(prev, None)] * rep)
            else:
                c2[16] += 1
                packed.append((16, rep - 3))
            rep = 0

    def flush_zero():
        nonlocal zrun
        if zrun:
            if zrun < 3:
                c2[0] += zrun
                packed.extend([(0, None)] * zrun)
            elif zrun <= 10:
                c2[17] += 1
                packed.append((17, zrun - 3))
            else:
                c2[18] += 1
                packed.append((18, zrun - 11))
            zrun = 0

    for cs in seq:
        if not cs:
            flush_rep()
            zrun += 1
            if zrun == 138:
                flush_zero()
        else:
            flush_zero()
            if cs != prev:
                flush_rep()
                c2[cs] += 1
                packed.append((cs, None))
            else:
                rep += 1
                if rep == 6:
                    flush_rep()
        prev = cs
    if rep:
        flush_rep()
    else:
        flush_zero()
    return packed, c2


def by_runs(seq):
    toks, c2 = [], [0] * 19
    i, n = 0, len(seq)
    while i < n:
        j = i
        while j < n and seq[j] == seq[i]:
            j += 1
        v, length = seq[i], j - i
        if v:
            rest = length - 1
            full, r = divmod(rest, 6)
            toks.append((v, None))
            toks.extend([(16, 3)] * full)
            toks.extend([(v, None)] * r if r < 3 else [(16, r - 3)])
            c2[v] += 1 + (r if r < 3 else 0)
            c2[16] += full + (1 if r >= 3 else 0)
        else:
            full, r = divmod(length, 138)
            toks.extend([(18, 127)] * full)
            if 0 < r < 3:
                toks.extend([(0, None)] * r)
                c2[0] += r
            elif 3 <= r <= 10:
                toks.append((17, r - 3))
                c2[17] += 1
            elif r > 10:
                toks.append((18, r - 11))
            c2[18] += full + (1 if r > 10 else 0)
        i = j
    return toks, c2


def test_closed_form_per_run_equals_the_state_machine():
    rng = random.Random(1)
    for _ in range(20000):
        n = rng.randint(1, 320)
        long_runs = rng.random() < 0.5
        seq = []
        while len(seq) < n:
            v = rng.choice([0, 0, 0, rng.randint(1, 12), rng.randint(1, 12)])
            length = rng.choice([1, 1, 2, 3, 4, 5, 6, 7, 8, 13, 14, 20, 137, 138, 139, 150, 280]) if long_runs else rng.randint(1, 12)
            seq += [v] * length
        seq = seq[:n]
        assert state_machine(seq) == by_runs(seq), seq
