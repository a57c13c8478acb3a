"""Atom selection queries — the subset of the groan selection language the reference's configurations and tests use
(/root/reference/src/analysis/common.rs:92-130 `create_group`, the queries of tests/tests_{aa,cg,ua}.rs):

    resname POPC POPE          name C22 C24 'C218'          name r'^C'        (regular expressions in r'...')
    resid 1 to 254  /  resid 7 12 30-40      serial 1 to 100          element name carbon  (elname) / element symbol C (elsymbol)
    @membrane @water @ion @protein (macros: lists of residue names)    NdxGroupName    all
    not X   X and Y   X or Y   ( ... )       also  !  &&  ||         molwith X   (whole molecules that contain an atom of X)

`select(structure, query, groups=None)` -> boolean mask over the atoms.  groan_rs (0.11.2, not in the checkout) owns the
real grammar; this evaluator restates its documented semantics for that subset: `and` binds tighter than `or`, `not`
tighter than both, keyword arguments are lists (any match), residue / atom numbers count from 1.
The @membrane macro there is a list of some 200 lipid residue names; the list here holds the common phospholipids,
sphingolipids, sterols and the Martini names of the same and can be extended through `MACROS`."""
import re
from typing import Dict, List, Optional, Sequence

import numpy as np

from .structure import Structure, guess_element

MACROS: Dict[str, Sequence[str]] = {
    "membrane": ("POPC POPE POPG POPS POPA POPI DOPC DOPE DOPG DOPS DOPA DPPC DPPE DPPG DPPS DMPC DMPE DMPG DMPS DLPC DLPE "
                 "DLPG DSPC DSPE DSPG DSPS DAPC DUPC DIPC DIPE PLPC PLPE PAPC PAPE PAPS SOPC SOPE SAPC SAPE SDPC SDPE "
                 "POSM PSM SSM DPSM DXSM PNSM CHOL CHL1 ERG DPG1 DPG3 DXG1 DXG3 PNG1 PNG3 XNG1 XNG3 DPCE DXCE PNCE XNCE "
                 "LPPC PIPC PIPE PIPI PIPS PUPC PUPE PUPI PUPS PEPC PRPC PRPE PRPS PQPE PQPS OPPC LSM BNSM DBSM TOCL CDL1 "
                 "CDL2 POP1 POP2 POP3 POP4 POP5 POP6 POP7 PAP1 PAP2 PAP3 PAP6 DOTAP DLIN").split(),
    "water": "SOL WAT HOH TIP3 TIP4 TIP5 SPC SPCE W WF PW H2O".split(),
    "ion": "NA CL K CA MG ZN NA+ CL- K+ CA2+ MG2+ SOD CLA POT CAL ION LI CS RB".split(),
    "protein": ("ALA ARG ASN ASP CYS GLN GLU GLY HIS HSD HSE HSP HID HIE HIP ILE LEU LYS MET PHE PRO SER THR TRP TYR VAL "
                "ASH GLH LYN CYX CYM ACE NME NH2").split(),
}
ELEMENT_NAMES = {"hydrogen": "H", "carbon": "C", "nitrogen": "N", "oxygen": "O", "phosphorus": "P", "sulfur": "S",
                 "sodium": "Na", "chlorine": "Cl", "potassium": "K", "calcium": "Ca", "magnesium": "Mg", "zinc": "Zn",
                 "fluorine": "F", "bromine": "Br", "iodine": "I", "lithium": "Li", "iron": "Fe"}
_KEYWORDS = {"resname", "name", "atomname", "resid", "serial", "element", "elname", "elsymbol", "molwith", "molecule",
             "and", "or", "not", "to", "all", "(", ")", "&&", "||", "!"}
_TOKEN = re.compile(r"""\s*(r'[^']*'|r"[^"]*"|'[^']*'|"[^"]*"|&&|\|\||[()!]|[^\s()!&|]+)""")


class SelectError(ValueError):
    pass


def _tokens(query: str) -> List[str]:
    out, pos = [], 0
    while pos < len(query):
        m = _TOKEN.match(query, pos)
        if not m:
            if query[pos:].strip() == "":
                break
            raise SelectError(f"cannot read the query at '{query[pos:]}'")
        out.append(m.group(1))
        pos = m.end()
    return out


class _Parser:
    def __init__(self, structure: Structure, groups: Optional[Dict[str, np.ndarray]]):
        self.s = structure
        self.groups = groups or {}
        self.n = structure.n_atoms
        self.names = np.array(structure.names)
        self.resnames = np.array(structure.resnames)
        self.resids = np.asarray(structure.resids)
        self._elements = None
        self.t: List[str] = []
        self.i = 0

    # ---- grammar: or_expr := and_expr (or and_expr)* ; and_expr := unary (and unary)* ; unary := not unary | atom
    def parse(self, query: str) -> np.ndarray:
        self.t, self.i = _tokens(query), 0
        if not self.t:
            raise SelectError("empty query")
        mask = self.or_expr()
        if self.i != len(self.t):
            raise SelectError(f"unexpected '{self.t[self.i]}' in the query")
        return mask

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else None

    def take(self):
        tok = self.peek()
        self.i += 1
        return tok

    def or_expr(self):
        m = self.and_expr()
        while self.peek() in ("or", "||"):
            self.take()
            m = m | self.and_expr()
        return m

    def and_expr(self):
        m = self.unary()
        while self.peek() in ("and", "&&"):
            self.take()
            m = m & self.unary()
        return m

    def unary(self):
        tok = self.peek()
        if tok in ("not", "!"):
            self.take()
            return ~self.unary()
        if tok in ("molwith", "molecule"):
            self.take()
            if self.peek() == "with":
                self.take()
            return self.molwith(self.unary())
        return self.atom()

    def args(self) -> List[str]:
        out = []
        while self.peek() is not None and self.peek() not in _KEYWORDS:
            out.append(self.take())
        if not out:
            raise SelectError("a keyword of the query has no arguments")
        return out

    def atom(self):
        tok = self.take()
        if tok is None:
            raise SelectError("the query ends too early")
        if tok == "(":
            m = self.or_expr()
            if self.take() != ")":
                raise SelectError("missing ')' in the query")
            return m
        if tok == "all":
            return np.ones(self.n, dtype=bool)
        if tok == "resname":
            return self.match(self.resnames, self.args())
        if tok in ("name", "atomname"):
            return self.match(self.names, self.args())
        if tok == "resid":
            return self.numbers(self.resids)
        if tok == "serial":
            return self.numbers(np.arange(1, self.n + 1))
        if tok in ("element", "elname", "elsymbol"):
            kind = tok
            if tok == "element":
                kind = {"name": "elname", "symbol": "elsymbol"}.get(self.take() or "", None)
                if kind is None:
                    raise SelectError("'element' must be followed by 'name' or 'symbol'")
            wanted = self.args()
            el = self.elements()
            if kind == "elname":
                unknown = [w for w in wanted if w.lower() not in ELEMENT_NAMES]
                if unknown:
                    raise SelectError(f"unknown element name '{unknown[0]}'")
                wanted = [w.lower() for w in wanted]
                return np.isin(el, wanted)
            symbols = {v.lower(): k for k, v in ELEMENT_NAMES.items()}
            return np.isin(el, [symbols.get(w.lower(), "?") for w in wanted])
        if tok.startswith("@"):
            if tok[1:] not in MACROS:
                raise SelectError(f"unknown macro '{tok}'")
            return np.isin(self.resnames, list(MACROS[tok[1:]]))
        name = tok.strip("'\"")
        if name in self.groups:
            g = np.asarray(self.groups[name])
            if g.dtype == bool:
                return g.copy()
            m = np.zeros(self.n, dtype=bool)
            m[g] = True
            return m
        raise SelectError(f"group '{name}' does not exist")

    def match(self, values: np.ndarray, wanted: List[str]) -> np.ndarray:
        m = np.zeros(self.n, dtype=bool)
        for w in wanted:
            if w[:2] in ("r'", 'r"'):
                try:
                    rx = re.compile(w[2:-1])
                except re.error as e:
                    raise SelectError(f"invalid regular expression {w}: {e}")
                m |= np.array([rx.search(v) is not None for v in values])     # groan: an unanchored match
            else:
                m |= values == w.strip("'\"")
        return m

    def numbers(self, values: np.ndarray) -> np.ndarray:
        m = np.zeros(self.n, dtype=bool)
        got = False
        while self.peek() is not None and self.peek() not in (_KEYWORDS - {"to"}):
            tok = self.take()
            lo = hi = None
            if re.fullmatch(r"\d+", tok):
                lo = hi = int(tok)
                if self.peek() in ("to", "-"):
                    self.take()
                    nxt = self.take()
                    if nxt is None or not re.fullmatch(r"\d+", nxt):
                        raise SelectError("a number range of the query has no end")
                    hi = int(nxt)
            elif re.fullmatch(r"\d+-\d+", tok):
                lo, hi = (int(x) for x in tok.split("-"))
            else:
                raise SelectError(f"'{tok}' is not a number")
            m |= (values >= lo) & (values <= hi)
            got = True
        if not got:
            raise SelectError("a keyword of the query has no arguments")
        return m

    def elements(self) -> np.ndarray:
        if self._elements is None:
            self._elements = np.array([guess_element(n, r) for n, r in zip(self.s.names, self.s.resnames)])
        return self._elements

    def molwith(self, mask: np.ndarray) -> np.ndarray:
        adj = self.s.bonds
        if adj is None:
            raise SelectError("'molwith' needs the bonds of the structure")
        out = np.zeros(self.n, dtype=bool)
        for a in np.flatnonzero(mask):
            if out[a]:
                continue
            stack = [int(a)]
            out[a] = True
            while stack:
                i = stack.pop()
                for j in adj[i]:
                    if not out[j]:
                        out[j] = True
                        stack.append(j)
        return out


def select(structure: Structure, query: str, groups: Optional[Dict[str, np.ndarray]] = None) -> np.ndarray:
    """Boolean mask of the atoms the query selects; `groups`: name -> index array or mask (NDX groups)."""
    return _Parser(structure, groups).parse(query)


def read_ndx(path: str) -> Dict[str, np.ndarray]:
    """GROMACS index file: `[ name ]` headers followed by 1-based atom numbers -> name -> 0-based index array."""
    groups: Dict[str, List[int]] = {}
    cur = None
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            if line.startswith("["):
                cur = line.strip("[] \t")
                groups.setdefault(cur, [])
            elif cur is not None:
                groups[cur] += [int(x) - 1 for x in line.split()]
    return {k: np.array(v, dtype=np.int64) for k, v in groups.items()}
