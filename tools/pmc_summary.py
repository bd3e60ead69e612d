"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/pmc_traffic.json.
FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section):
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read -> doubled here;
WRITE_SIZE is exact for streaming stores.  Per launch, averaged over the kernel's dispatches."""
import csv, glob, json, sys, collections

def load(d, name):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc

fe = load(sys.argv[1], "FETCH_SIZE"); wr = load(sys.argv[2], "WRITE_SIZE")
# (spmv: the solver's kernel -- k_spmv_cls on row classes since round 4; k_spmv_win<true, 8, false> is then the parity-CSR
#  product of the acceptance residual and is listed on its own)
classes = {"spmv": ("k_spmv_cls<", "k_spmv_pat<"), "spmv_parity_csr": "k_spmv_win<", "schwarz_apply": "k_apply", "assemble": ("k_assemble_pairs<", "k_assemble_tiles<"), "multidot": "k_multidot(",
           "multiaxpy": "k_multiaxpy(", "gs_dot": ("k_multidot2", "k_blockdot<"), "gs_update": ("k_axpy2", "k_blockaxpy<"), "gs_fused": "k_blockfuse<", "invert": "k_invert_reg<7"}
# (gs_dot / gs_update: the basis grows from launch to launch; the figure is the mean over all launches of the solve, like
# bench.py's byte model.  schwarz_apply with shared inverses is a gather kernel: the doubling of FETCH_SIZE is calibrated
# on streaming reads and may overstate its traffic)
out = {}
for key, pat in classes.items():
    pats = pat if isinstance(pat, tuple) else (pat,)
    fk = [v for k, vs in fe.items() if any(p in k for p in pats) for v in vs]
    wk = [v for k, vs in wr.items() if any(p in k for p in pats) for v in vs]
    if not fk: continue
    # drop gated-out launches (near-zero traffic) of the DGKS second pass
    fk2 = ([v for v in fk if v > 0.05 * max(fk)] or fk) if key in ("multidot", "multiaxpy") else fk
    wk2 = wk[:len(wk)]
    rd = 2.0 * 1024 * sum(fk2) / len(fk2)
    wrt = 1024 * sum(wk2) / max(1, len(wk2))
    names = sorted({k.replace("(anonymous namespace)::", "").split("(")[0] for k in fe if any(p in k for p in pats)})
    out[key] = {"kernel": ", ".join(names), "read_bytes_per_launch": rd, "write_bytes_per_launch": wrt, "hbm_bytes_per_launch": rd + wrt,
                "launches_sampled": len(fk2), "note": "FETCH_SIZE doubled (gfx950 128-B request correction), WRITE_SIZE as reported"}
flat = {k: v["hbm_bytes_per_launch"] for k, v in out.items()}
# the grid the passes were taken on (bench.py quotes `traffic` only for the same one): argv[4] = "214,214,214"
cells = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else None
# argv[5] / argv[6]: tag of the profile round and git SHA of the tree the passes ran on (bench.py stamps them next to `traffic`)
json.dump({"cells_per_gpu": cells, "n_gpus": 1, "tag": sys.argv[5] if len(sys.argv) > 5 else None,
           "git_sha": sys.argv[6] if len(sys.argv) > 6 else None, "detail": out, **flat}, open(sys.argv[3], "w"), indent=1)
print(json.dumps(flat, indent=1))
