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
