"""Per-launch HBM traffic of ONE contour step from the two rocprofv3 counter passes of bench.py (FETCH_SIZE, WRITE_SIZE in
separate runs: tools/collect_profiles_r03.sh).  FETCH_SIZE is doubled (gfx950 reports half of the bytes of wide coalesced reads,
MI355X_MICROARCH.md), WRITE_SIZE is exact.  usage: python tools/pmc_per_launch.py <dir_fetch> <dir_write>"""
import collections, csv, glob, re, sys


def load(d, counter):
    rows = []
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


def short(n):
    m = re.search(r'(k_[a-z_0-9]+)(<[^>]*>)?', n)
    if not m:
        return n[:28]
    t = m.group(2) or ""
    t = t.replace("HIP_vector_type<double, 2u>", "c128").replace("HIP_vector_type<float, 2u>", "c64")
    return m.group(1) + t


def one_step(rows, scale):
    names = [short(r["Kernel_Name"]) for r in rows]
    starts = [i for i, n in enumerate(names) if n.startswith("k_gather_rhs")]
    a, b = starts[-2], starts[-1]          # a whole iteration: contour step + the Rayleigh-Ritz step behind it
    return [(names[i], float(rows[i]["Counter_Value"]) * scale) for i in range(a, b)]


F = one_step(load(sys.argv[1], "FETCH_SIZE"), 2 * 1024.0)
W = one_step(load(sys.argv[2], "WRITE_SIZE"), 1024.0)
tot = collections.OrderedDict()
for (n, fb), (_, wb) in zip(F, W):
    t = tot.setdefault(n, [0, 0.0, 0.0]); t[0] += 1; t[1] += fb; t[2] += wb
print("one FEAST iteration of C3 (8 units = 16 nodes, m0 = 64), HBM bytes per kernel: launches, GB read, GB written")
for n, (c, fb, wb) in tot.items():
    print(f"  {n:34s} {c:4d} {fb / 1e9:8.2f} {wb / 1e9:8.2f}")
print("launch sequence of the contour step (forward sweep leaves -> root, backward root -> leaves): GB read, GB written")
for j, ((n, fb), (_, wb)) in enumerate(zip(F, W)):
    if n.startswith(("k_apply2", "k_fwd_assemble", "k_splitk", "k_gather_rhs", "k_accumulate")):
        print(f"  {j:3d} {n:34s} {fb / 1e9:7.3f} {wb / 1e9:7.3f}")
