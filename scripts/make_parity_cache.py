"""Oracle results for scripts/parity_survey.py (run in the build container, ~1 min per 256 items on 8 cores): N bench
items from `first` on (2 qubits, Pauli or SIC in-basis, trace preserving or not), fixed-100 and converge mode, Choi
matrices and counters -> scripts/cache/oracle_<basis>[_tni][_<first>].npz.
usage: python scripts/make_parity_cache.py [pauli|sic|sic3] [N] [first] [tni]      (sic3: 3 qubits, SIC in-basis)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "forest-benchmarking_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ["OMP_NUM_THREADS"] = "1"
import numpy as np
from multiprocessing import Pool
from fbx import synthetic
BASIS = sys.argv[1] if len(sys.argv) > 1 else "pauli"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
FIRST = int(sys.argv[3]) if len(sys.argv) > 3 else 0
TP = not (len(sys.argv) > 4 and sys.argv[4] == "tni")
NQ = 3 if BASIS.endswith("3") else 2
design, us, e, c = synthetic.process_batch(NQ, BASIS.rstrip("3"), N, first_item=FIRST)


def work(b):
    from fbx_oracle import design as od, estimators as oe
    d = od.Design(NQ, "process", design.in_labels, design.paulis, design.coefs)
    A = oe.design_matrix_A(d)
    x, s = oe.pgdb_process_estimate(d, e[b], c[b], trace_preserving=TP, mode="fixed", max_iters=100, A=A, return_stats=True)
    y, t = oe.pgdb_process_estimate(d, e[b], c[b], trace_preserving=TP, A=A, return_stats=True)
    return x, s["backtracks"], s["dykstra"], y, t["backtracks"], t["dykstra"], t["iterations"]


if __name__ == "__main__":
    with Pool(os.cpu_count() if NQ == 2 else 6) as p:
        res = p.map(work, range(N))
    os.makedirs(os.path.join(ROOT, "scripts", "cache"), exist_ok=True)
    name = "oracle_" + BASIS + ("" if TP else "_tni") + (f"_{FIRST}" if FIRST else "") + ".npz"
    np.savez(os.path.join(ROOT, "scripts", "cache", name), basis=BASIS, first=FIRST, tp=TP,
             fixed=np.array([r[0] for r in res]), fixed_bt=np.array([r[1] for r in res]), fixed_dyk=np.array([r[2] for r in res]),
             conv=np.array([r[3] for r in res]), conv_bt=np.array([r[4] for r in res]), conv_dyk=np.array([r[5] for r in res]),
             conv_it=np.array([r[6] for r in res]))
    print("done", name, N)
