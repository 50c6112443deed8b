"""The inverse one-step neighbourhood behind the offset filter of K-BC1 and the item filter of K-BC2 (csrc/smi_bc.hip: n1_member,
k_set_nb / k_set_n1) against the reference's own mutation operators as tests/pymodel.py restates them from the bytecode
(NucleicAcidTwoBitPerBase.getLongHashReplaceByteDeg / InsertByteDeg / deleteByte).

Claim the kernels rely on: if ANY sequence of the reference's one-step enumeration of a 16-mer K -- K itself, its 48 substitutions, the
60 insertions behind positions 0..14, the 15 deletions of positions 0..14 with any appended base -- is a barcode w, then K is one of the
169 sequences n1_member(w, 0..168).  So a K outside the union of those sets over the barcode list has no match at edit distance <= 1 and
its 124 probes need not be made.  Checked here in plain Python on random and degenerate 16-mers for every one of the mutants, and the
second claim (K-BC2: when K is itself a barcode, each of its children except the one that lost position 0 has K in ITS neighbourhood
set) as well.  The GPU suites then check whole results with and without the filters (SMI_BC1_NO_FILTER / SMI_BC2_NO_FILTER)."""
import random

import pymodel as pm

N = 16
MASK32 = (1 << 32) - 1


def lowmask(nbits):
    return MASK32 if nbits >= 32 else (1 << nbits) - 1


def n1_member(k, slot):
    """the restatement of n1_member in smi_bc.hip"""
    if slot < 48:
        return k ^ ((slot % 3 + 1) << (30 - 2 * (slot // 3)))
    if slot < 108:
        j = slot - 48
        p1 = 1 + j // 4
        sh = 30 - 2 * p1
        return (k & ~lowmask(sh + 2) & MASK32) | ((k & lowmask(sh)) << 2) | (j & 3)
    if slot < 168:
        j = slot - 108
        q = j // 4
        top = 32 - 2 * q
        return (k & ~lowmask(top) & MASK32) | ((j & 3) << (30 - 2 * q)) | ((k & lowmask(top)) >> 2)
    return k


def neighbourhood(w):
    return {n1_member(w, s) for s in range(169)}


def forward_mutants(k):
    """every sequence the reference's level-1 enumeration of K can probe, as (kind, position, value); 64-bit values whose upper half is
    not zero (insert behind position 14 of a K that does not end in A) can never equal a barcode and are left out, as in the kernels"""
    out = [("exact", -1, k)]
    for pos in range(N):
        cur = (k >> (2 * (N - 1 - pos))) & 3
        for b, v in enumerate(pm.replace_deg(k, pos, N)):
            if b != cur:
                out.append(("sub", pos, v))
        if pos < N - 1:
            for v in pm.insert_deg(k, pos, N):
                if v >> 32 == 0:
                    out.append(("ins", pos, v))
            for base4 in (1, 2, 4, 8):  # whatever read base is appended
                out.append(("del", pos, pm.delete_byte(k, base4, pos, N)))
    return out


def keys_for_test(rng, n):
    ks = [rng.getrandbits(32) for _ in range(n)]
    ks += [0, MASK32, 0x55555555, 0xAAAAAAAA, 0x0000FFFF, 0xFFFF0000, 0x01234567, 0x33333333, 0xCCCCCCCC, 3, 3 << 30, 1 << 31]
    # short repeats: where different mutations coincide
    ks += [int("".join(rng.choice(["00", "01", "10", "11"]) * 2 for _ in range(8)), 2) for _ in range(50)]
    return ks


def test_forward_mutants_lie_in_the_inverse_neighbourhood():
    rng = random.Random(20261003)
    for k in keys_for_test(rng, 300):
        for kind, pos, m in forward_mutants(k):
            assert m >> 32 == 0
            assert k in neighbourhood(m), (hex(k), kind, pos, hex(m))


def test_children_of_a_barcode_have_it_in_their_neighbourhood():
    """K-BC2's second table: K a barcode, X a child of K (not the deletion of position 0) => X is in neighbourhood(K)"""
    rng = random.Random(7)
    for k in keys_for_test(rng, 300):
        nb = neighbourhood(k)
        for kind, pos, x in forward_mutants(k):
            if kind == "del" and pos == 0:
                continue
            assert x in nb, (hex(k), kind, pos, hex(x))


def test_neighbourhood_is_not_everything():
    """the filter filters: 169 members per barcode at most, and a random 16-mer is in the neighbourhood of a random one with
    probability ~ 169 / 2^32"""
    rng = random.Random(3)
    w = rng.getrandbits(32)
    nb = neighbourhood(w)
    assert 100 < len(nb) <= 169
    assert sum(1 for _ in range(20000) if rng.getrandbits(32) in nb) == 0
