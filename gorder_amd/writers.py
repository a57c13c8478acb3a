"""Text output in the reference's layouts, from the result tree of `structure.results_tree[_ua]`:
YAML (presentation/yaml_presenter.rs:80-136: nested mapping, values rounded to 4 decimals, NaN as `.nan`) and CSV
(presentation/csv_presenter.rs: one line per heavy atom or coarse-grained bond, fixed 4 decimals, NaN as `NaN`, empty
fields for hydrogens an atom does not have).  Both can be compared with the reference's files by its own rule
(tests/common/mod.rs:95-150: same items line by line, numbers within 2e-4)."""
import re
from typing import List, Optional


def _num(x) -> str:
    if x != x:
        return ".nan"
    s = repr(float(x))
    return s[:-2] if s.endswith(".0") else s


def yaml_text(tree: dict, header: Optional[str] = None) -> str:
    out: List[str] = [header] if header else []

    def emit(node, indent):
        pad = "  " * indent
        for key, val in node.items():
            if isinstance(val, dict):
                out.append(f"{pad}{key}:")
                emit(val, indent + 1)
            elif isinstance(val, (list, tuple)):       # united atoms: `bonds:` is a sequence of mappings
                out.append(f"{pad}{key}:")
                for item in val:
                    first = True
                    for k2, v2 in item.items():
                        lead = f"{pad}- " if first else f"{pad}  "
                        first = False
                        if isinstance(v2, dict):
                            out.append(f"{lead}{k2}:")
                            emit(v2, indent + 2)
                        else:
                            out.append(f"{lead}{k2}: {_num(v2)}")
            else:
                out.append(f"{pad}{key}: {_num(val)}")

    emit(tree, 0)
    return "\n".join(out) + "\n"


_ATOM_KEY = re.compile(r"^(\S+) (\S+) \((\d+)\)$")
_BOND_KEY = re.compile(r"^(\S+) (\S+) \((\d+)\) - (\S+) (\S+) \((\d+)\)$")


def _fixed(x) -> str:
    return "NaN" if x != x else f"{x:.4f}"


def _cells(value: dict, which: List[str], errors: bool) -> List[str]:
    """One order parameter as CSV cells: per leaflet slot the mean (and its error)."""
    cells = []
    for w in which:
        v = value[w]
        if errors:
            cells += [_fixed(v["mean"]), _fixed(v["error"])]
        else:
            cells.append(_fixed(v))
    return cells


def csv_text(tree: dict) -> str:
    """AA / UA: molecule,residue,atom,relative index,total…,hydrogen #k…   CG: molecule,atom 1,atom 2,…"""
    molecules = [(k, v) for k, v in tree.items() if k != "average order"]
    sample = tree["average order"]
    which = [w for w in ("total", "upper", "lower") if w in sample]
    leaflets = len(which) == 3
    errors = isinstance(sample["total"], dict)
    first_op = next(iter(molecules[0][1]["order parameters"].values()))
    atom_based = "bonds" in first_op
    suffix = {"total": " full membrane", "upper": " upper leaflet", "lower": " lower leaflet"}

    def columns(name: str) -> List[str]:
        cols = []
        for w in which:
            label = (name + suffix[w]) if leaflets else name
            cols.append(label)
            if errors:
                cols.append(label + " error")
        return cols

    lines = []
    if atom_based:
        n_h = 0
        for _, mol in molecules:
            for entry in mol["order parameters"].values():
                n_h = max(n_h, len(entry["bonds"]))
        head = ["molecule", "residue", "atom", "relative index"] + columns("total")
        for k in range(n_h):
            head += columns(f"hydrogen #{k + 1}")
        lines.append(",".join(head))
        width = len(columns("x"))
        for mname, mol in molecules:
            for key, entry in mol["order parameters"].items():
                res, atom, rel = _ATOM_KEY.match(key).groups()
                row = [mname, res, atom, rel] + _cells(entry, which, errors)
                bonds = entry["bonds"]
                bonds = list(bonds.values()) if isinstance(bonds, dict) else list(bonds)
                for k in range(n_h):
                    row += _cells(bonds[k], which, errors) if k < len(bonds) else [""] * width
                lines.append(",".join(row))
    else:
        def cg_columns():
            cols = []
            for w in which:
                label = suffix[w].strip()
                cols.append(label)
                if errors:
                    cols.append(label + " error")
            return cols
        lines.append(",".join(["molecule", "atom 1", "atom 2"] + cg_columns()))
        for mname, mol in molecules:
            for key, entry in mol["order parameters"].items():
                _, a1, _, _, a2, _ = _BOND_KEY.match(key).groups()
                lines.append(",".join([mname, a1, a2] + _cells(entry, which, errors)))
    return "\n".join(lines) + "\n"


def _pm(value, errors: bool) -> str:
    if errors:
        if value["mean"] != value["mean"]:             # tab_presenter.rs:135-139: one centred NaN, no error beside it
            return f"{'NaN':^17s}"
        return f"{_fixed(value['mean']):>8s} ± {_fixed(value['error'])}"
    return f"{_fixed(value):>8s}"


def tab_text(tree: dict, header: Optional[str] = None) -> str:
    """The table layout (presentation/tab_presenter.rs): per molecule type one row per heavy atom (TOTAL and the
    hydrogens) or per coarse-grained bond, an AVERAGE row, and the average of all molecule types at the end.  The
    reference compares these files token by token (tests/common/mod.rs:113-124), so column widths are cosmetic."""
    molecules = [(k, v) for k, v in tree.items() if k != "average order"]
    sample = tree["average order"]
    which = [w for w in ("total", "upper", "lower") if w in sample]
    leaflets = len(which) == 3
    errors = isinstance(sample["total"], dict)
    atom_based = "bonds" in next(iter(molecules[0][1]["order parameters"].values()))
    out = [header or "# order parameters", ""]
    slot_names = "      ".join(("FULL", "UPPER", "LOWER")) if leaflets else None

    def cell(value) -> str:
        return "   ".join(_pm(value[w], errors) for w in which)

    def head_rows(groups: List[str], label: str):
        if atom_based and not (leaflets and groups == ["TOTAL"] and label == "all"):
            out.append(" " * 10 + "  |  ".join(f"{g:^{len(cell(sample))}s}" for g in groups) + "  |")
        if leaflets:
            out.append(" " * 10 + "  |  ".join(slot_names for _ in groups) + "  |")
        elif not atom_based:
            out.append(" " * 18 + "FULL   |")

    for mname, mol in molecules:
        out.append(f"Molecule type {mname}")
        ops = mol["order parameters"]
        if atom_based:
            n_h = max(len(e["bonds"]) for e in ops.values())
            hname = "HYDROGEN" if (leaflets or errors) else "H"      # the narrow table abbreviates
            head_rows(["TOTAL"] + [f"{hname} #{k + 1}" for k in range(n_h)], "mol")
            for key, entry in ops.items():
                atom = _ATOM_KEY.match(key).group(2)
                bonds = entry["bonds"]
                bonds = list(bonds.values()) if isinstance(bonds, dict) else list(bonds)
                cells = [cell(entry)] + [cell(bonds[k]) if k < len(bonds) else " " * len(cell(entry)) for k in range(n_h)]
                out.append(f"{atom:<8s}" + "  |  ".join(cells) + "  |")
        else:
            head_rows(["FULL"], "mol")
            for key, entry in ops.items():
                _, a1, _, _, a2, _ = _BOND_KEY.match(key).groups()
                out.append(f"{a1 + ' - ' + a2:<15s}" + cell(entry) + "  |")
        out.append(f"{'AVERAGE':<8s}" + cell(mol["average order"]) + "  |")
        out.append("")
    out.append("All molecule types")
    head_rows(["TOTAL"] if atom_based else ["FULL"], "all")
    out.append(f"{'AVERAGE':<8s}" + cell(sample) + "  |")
    return "\n".join(out) + "\n"


def xvg_text(tree: dict, molecule: str, header: Optional[str] = None, united: bool = False) -> str:
    """One molecule type as an xvg data set (presentation/xvg_presenter.rs): a numbered line per heavy atom / bond,
    full membrane and, with leaflets, upper and lower."""
    mol = tree[molecule]
    ops = mol["order parameters"]
    sample = tree["average order"]
    which = [w for w in ("total", "upper", "lower") if w in sample]
    errors = isinstance(sample["total"], dict)
    atom_based = "bonds" in next(iter(ops.values()))
    out = [header or "# order parameters",
           f'@    title "{("United-atom" if united else "Atomistic") if atom_based else "Coarse-grained"} order parameters for molecule type {molecule}"',
           f'@    xaxis label "{"Atom" if atom_based else "Bond"}"',
           f'@    yaxis label "{"-Sch" if atom_based else "S"}"']
    legends = {"total": "Full membrane", "upper": "Upper leaflet", "lower": "Lower leaflet"}
    for k, w in enumerate(which):
        out.append(f'@    s{k} legend "{legends[w]}"')
    out.append("@TYPE xy")
    for n, (key, entry) in enumerate(ops.items(), start=1):
        if atom_based:
            out.append(f"# Atom {_ATOM_KEY.match(key).group(2)}:")
        else:
            _, a1, _, _, a2, _ = _BOND_KEY.match(key).groups()
            out.append(f"# Bond {a1} - {a2}:")
        vals = [entry[w]["mean"] if errors else entry[w] for w in which]
        out.append(f"{n:<4d} " + " ".join(f"{_fixed(v):>8s}" for v in vals) + " ")
    return "\n".join(out) + "\n"


def convergence_text(timewise, labels, analysis: str, leaflets: bool, header: Optional[str] = None, step: int = 1) -> str:
    """Convergence of the molecule types' average order parameters (TimeWiseData::prefix_average, timewise.rs:259-274;
    presentation/convergence.rs): line n holds, per molecule type (and leaflet), the average over the first n analysed
    frames — cumulative tick sum / cumulative sample count by the truncating integer division, sign as in the
    other outputs.  `timewise` = (sums, counts) [frames][3][n_acc] as returned by the engines; `labels` as from
    build_tables*; x = the frame's number in the trajectory, 1 + n * step."""
    import numpy as np
    sums, counts = (np.asarray(x) for x in timewise)
    sign = -1.0 if analysis in ("aa", "ua") else 1.0
    which = ["full", "upper", "lower"] if leaflets else [""]
    out = [header or "# order parameters",
           '@    title "Convergence of average order parameters for individual molecule types"',
           '@    xaxis label "Frame number"',
           f'@    yaxis label "{"-Sch" if analysis in ("aa", "ua") else "S"}"']
    cols = []
    for ml in labels:
        n_slots = sum(c.n_h for c in ml.carbons) if hasattr(ml, "carbons") else len(ml.bonds)
        sl = slice(ml.slot0, ml.slot0 + n_slots)
        for w, name in enumerate(which):
            out.append(f'@    s{len(cols)} legend "{(ml.name + " " + name).strip()}"')
            cs = np.cumsum(sums[:, w, sl].sum(axis=1).astype(np.int64))
            cn = np.cumsum(counts[:, w, sl].sum(axis=1).astype(np.int64))
            col = []
            for s_, n_ in zip(cs, cn):
                if n_ == 0:
                    col.append(float("nan"))
                else:
                    q = abs(int(s_)) // int(n_)
                    col.append(sign * float(np.float32((-q if s_ < 0 else q) / 1e6)))
            cols.append(col)
    out.append("@TYPE xy")
    for f in range(sums.shape[0]):
        out.append(f"{f * step + 1:<4d} " + " ".join(f"{_fixed(c[f]):>8s}" for c in cols) + " ")
    return "\n".join(out) + "\n"
