"""Text-level mutations of an MLIP-3 .mtp file that keep it valid but leave the shapes real files have: they exercise
the corners of the native product schedule (lammps_mtp_kokkos_amd/csrc/mtp_potential.cpp, finalize) against the
reference's in-order semantics (pair_mtp.cpp:196-233), which the oracle follows row by row."""
import re


def _block(text, name):
    m = re.search(r"(%s\s*=\s*\{)(.*?)(\}\s*\n)(?=\s*[a-z_]+\s*=|\s*$)" % name, text, re.S)
    assert m, name
    return m


def mutate_mtp(src, dst, late_writer=True, dup_mapping=True):
    """late_writer: the last row that adds to a stored product X which rows into never-read scalars use as a factor is
    moved to the end of alpha_index_times -- those scalars then see a partial X in file order, so their rows cannot be
    deferred (they must lose the leaf treatment).  dup_mapping: the last scalar is mapped onto the moment of the first
    never-read scalar -- two coefficients on one moment: both count in the energy, the last one seeds the adjoint
    (pair_mtp.cpp:204-218).  Returns a dict of what was done."""
    text = open(src).read()
    head, tail = text, b""
    mt = _block(text, "alpha_index_times")
    rows = [tuple(int(v) for v in r) for r in re.findall(r"\{\s*(-?\d+)\s*,\s*(-?\d+)\s*,\s*(-?\d+)\s*,\s*(-?\d+)\s*\}", mt.group(2))]
    mm = _block(text, "alpha_moment_mapping")
    mapping = [int(v) for v in re.findall(r"-?\d+", mm.group(2))]
    factors = set()
    for a, b, _, _ in rows:
        factors.add(a)
        factors.add(b)
    targets = set(t for *_, t in rows)
    leaves = [t for t in sorted(targets) if t not in factors]
    info = {"rows": len(rows), "leaves": len(leaves)}
    if late_writer:
        pick = None
        for k, (a, b, _, t) in enumerate(rows):
            if t in leaves:
                for x in (a, b):
                    writers = [j for j, r in enumerate(rows) if r[3] == x]
                    if writers and max(writers) < k:
                        pick = (max(writers), x)
                        break
            if pick:
                break
        assert pick, "no stored product feeds a leaf row"
        row = rows.pop(pick[0])
        rows.append(row)
        info["moved_row"], info["moved_target"] = pick
    if dup_mapping:
        first_leaf_scalar = next(m for m in mapping if m in leaves)
        mapping[-1] = first_leaf_scalar
        info["dup_moment"] = first_leaf_scalar
    new_rows = ", ".join("{%d, %d, %d, %d}" % r for r in rows)
    text = text[:mt.start(2)] + new_rows + text[mt.end(2):]
    mm = _block(text, "alpha_moment_mapping")
    text = text[:mm.start(2)] + ", ".join(str(v) for v in mapping) + text[mm.end(2):]
    open(dst, "w").write(text)
    return info
