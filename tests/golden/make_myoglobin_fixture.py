#!/usr/bin/env python3
"""Build tests/golden/myoglobin.npz from the reference's committed Myoglobin example.

Run in the build container only (needs /root/reference):
    python tests/golden/make_myoglobin_fixture.py

The output is DATA: the inputs of the reference's benchmark run
(examples/Data/Proteins/Myoglobin/{Myoglobin.aln.sel.mase,Myo.dnd}, options
examples/Proteins/Benchmark/CoMap/comap.bpp:7-80) and the outputs the reference committed for
it (examples/Proteins/Benchmark/CoMap/Myo_*.vec, Myo.infos).  No reference source is copied.
"""
import os
import sys
import numpy as np

REF = "/root/reference/examples"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle.np_oracle import parse_newick  # noqa: E402

AA = "ARNDCQEGHILKMFPSTWYV"
# ambiguity codes of Bio++'s ProteicAlphabet that occur in the data: B={D,N}, Z={E,Q}, X=any
AMBIG = {"B": "DN", "Z": "EQ", "X": AA, "J": "IL"}


def read_mase(path):
    names, seqs = [], []
    cur = None
    with open(path) as fh:
        for line in fh:
            line = line.rstrip("\n")
            if line.startswith(";"):
                cur = None
                continue
            if cur is None:
                names.append(line.strip())
                seqs.append([])
                cur = seqs[-1]
            else:
                cur.append(line.strip())
    return names, ["".join(s) for s in seqs]


def read_vec(path):
    with open(path) as fh:
        header = fh.readline().rstrip("\n").split("\t")
        coords = np.array([int(h[4:]) for h in header[2:]], dtype=np.int32)
        ids, bl, rows = [], [], []
        for line in fh:
            f = line.rstrip("\n").split("\t")
            if len(f) < 3:
                continue
            ids.append(int(f[0]))
            bl.append(float(f[1]))
            rows.append([float(x) for x in f[2:]])
    return coords, np.array(ids), np.array(bl), np.array(rows)


def main():
    names, seqs = read_mase(f"{REF}/Data/Proteins/Myoglobin/Myoglobin.aln.sel.mase")
    L = len(seqs[0])
    assert all(len(s) == L for s in seqs)
    T = len(seqs)
    raw = np.array([list(s.upper()) for s in seqs])
    # input.sequence.sites_to_use = nogap (comap.bpp:17)
    nogap = [j for j in range(L) if "-" not in raw[:, j]]
    # input.remove_const = yes (comap.bpp:25; CoETools.cpp:347-355, SiteTools::isConstant(site, ignoreUnknown=true))
    keep = []
    for j in nogap:
        col = [c for c in raw[:, j] if c != "X"]
        if len(set(col)) > 1:
            keep.append(j)
    coords = np.array([j + 1 for j in keep], dtype=np.int32)

    # codes: 0..19 states, 20.. ambiguity ids
    ambig_syms = sorted(AMBIG)
    code_of = {a: i for i, a in enumerate(AA)}
    masks = [1 << i for i in range(20)]
    for s in ambig_syms:
        code_of[s] = len(masks)
        masks.append(sum(1 << AA.index(a) for a in AMBIG[s]))
    aln = np.array([[code_of[raw[t, j]] for j in keep] for t in range(T)], dtype=np.uint8)

    with open(f"{REF}/Data/Proteins/Myoglobin/Myo.dnd") as fh:
        parent, blen, node_names = parse_newick(fh.read())
    leaf_of_taxon = np.array([node_names.index(n) for n in names], dtype=np.int32)

    out = dict(aln=aln, masks=np.array(masks, dtype=np.uint32), coords=coords, parent=parent, blen=blen,
               leaf_of_taxon=leaf_of_taxon, taxa=np.array(names), alpha=np.float64(0.985435), ncat=np.int32(4))
    bench = f"{REF}/Proteins/Benchmark/CoMap"
    for tag in ["unif", "decomp", "naive", "laplace", "unif_grantham", "decomp_grantham", "naive_grantham"]:
        c, ids, bl, rows = read_vec(f"{bench}/Myo_{tag}.vec")
        assert np.array_equal(c, coords), tag
        assert np.array_equal(ids, np.arange(len(ids)))
        out[f"vec_{tag}"] = rows           # [B, N]
        out["vec_blen"] = bl
    rc, pr, ll, cs = [], [], [], []
    with open(f"{bench}/Myo.infos") as fh:
        hdr = fh.readline().split()
        assert hdr == ["Group", "IsComplete", "IsConstant", "RC", "PR", "logLn"], hdr
        for line in fh:
            f = line.split()
            cs.append(int(f[0].strip("[]")))
            rc.append(int(f[3]))
            pr.append(float(f[4]))
            ll.append(float(f[5]))
    assert np.array_equal(np.array(cs), coords)
    out.update(infos_rc=np.array(rc, dtype=np.int32), infos_pr=np.array(pr), infos_logl=np.array(ll))
    # the first lines of the committed text outputs, verbatim (DATA): pin the .vec / .infos writers' formatting
    with open(f"{bench}/Myo_unif.vec") as fh:
        out["vec_unif_text_head"] = np.array("".join([fh.readline() for _ in range(4)]))
    with open(f"{bench}/Myo.infos") as fh:
        out["infos_text_head"] = np.array("".join([fh.readline() for _ in range(6)]))
    np.savez_compressed(os.path.join(HERE, "myoglobin.npz"), **out)
    print("sites", len(keep), "taxa", T, "nodes", len(parent), "->", os.path.join(HERE, "myoglobin.npz"))


if __name__ == "__main__":
    main()
