"""Read the stderr of scripts/pgdb1_phase_profile.py: per configuration (and per launch of the binned path) the share of wave
cycles per phase and the trip counts a wavefront runs against the mean over its lanes."""
import sys
rows = []
def show(tag, z):
    cyc = z[0:4]; tot = sum(cyc) or 1
    wo, lo = max(z[5], 1), max(z[6], 1)
    print(f"  {tag}: wave-iterations {z[5]:8d} lanes/wave {lo / wo:5.1f} | kcycles per wave-iteration {tot / wo / 1e3:7.1f} (kernel total per wave-iter {z[4] / wo / 1e3:7.1f}) | grad {cyc[0] / tot:5.1%} proj {cyc[1] / tot:5.1%} upd+cost {cyc[2] / tot:5.1%} line {cyc[3] / tot:5.1%} | "
          f"kernel: load {z[16] / wo / 1e3:5.1f}k compute {z[17] / wo / 1e3:6.1f}k outputs {z[18] / wo / 1e3:5.1f}k atomics {z[19] / wo / 1e3:5.1f}k stores {z[20] / wo / 1e3:5.1f}k | "
          f"Dykstra {z[7] / wo:6.2f} (lane {z[8] / lo:5.2f}) sweeps/Dyk {z[9] / max(z[7], 1):4.2f} (lane {z[10] / max(z[8], 1):4.2f}) | halvings {z[11] / wo:5.2f} (lane {z[12] / lo:5.2f}) prep {z[13] / wo:4.2f} full {z[14] / wo:5.2f} (lane {z[15] / lo:5.3f})")
for line in open(sys.argv[1]):
    if line.startswith("P1PROF"):
        v = [int(x) for x in line.split()[1:]]
        rows.append((v[0], v[1:]))
    elif line.startswith("##"):
        print(line.strip())
        if rows:
            tot = [sum(r[1][k] for r in rows) for k in range(24)]
            show("all  ", tot)
            if len(rows) > 1:
                for s, z in rows:
                    show(f"st{s:3d}", z)
        rows = []
