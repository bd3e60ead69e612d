#!/usr/bin/env python
"""Reads the NUMERIC TABLES of the reference as data and writes them to tests/golden/ref_tables.json:

  * simplex quadrature rules (points, weights, the degree remapping) from FE::getQuadratureValues,
    feddlib/core/FE/FE_def.hpp (dim 2 and the 'P' branch of dim 3);
  * reference basis functions and gradients (P1, P2; 2D, 3D) from FE::phi / FE::gradPhi, evaluated at every
    quadrature point of those rules and at the reference element's nodes;
  * the structured generators' element tables: the two triangles of a square cell and the six tetrahedra of a cube
    cell as offsets into the cell's corners (MeshStructured::buildMesh2D / buildMesh3D, P1 branches).

Run in the build container (it reads /root/reference; nothing at test time does):
    python tests/golden/make_ref_tables.py
The file it writes holds numbers only (no source text).  tests/test_ref_tables.py compares the product's tables
(libfedd_hip.so, through fedd_fe_tables / the structured generator) and the oracle's against it, which removes the
risk that product and oracle share one transcription error."""
import json
import math
import os
import re
import sys

REF = os.environ.get("FEDD_REFERENCE", "/root/reference")
FE_DEF = os.path.join(REF, "feddlib/core/FE/FE_def.hpp")
MESH_DEF = os.path.join(REF, "feddlib/core/Mesh/MeshStructured_def.hpp")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_tables.json")


def function_body(text, signature_regex):
    """text of the first function whose header matches (from its opening brace to the matching one)"""
    m = re.search(signature_regex, text)
    assert m, signature_regex
    i = text.index("{", m.end())
    depth, j = 0, i
    while True:
        if text[j] == "{":
            depth += 1
        elif text[j] == "}":
            depth -= 1
            if depth == 0:
                return text[i:j + 1]
        j += 1


TOKEN = re.compile(r"""
      (?P<sw>switch\s*\(\s*(?P<swvar>\w+)\s*\))
    | (?P<case>case\s+(?P<caseval>\d+)\s*:)
    | (?P<default>default\s*:)
    | (?P<if>(?:else\s+)?if\s*\((?P<cond>[^{;]*?)\)\s*(?=\{|\n|[A-Za-z]))
    | (?P<else>else\s*(?=\{))
    | (?P<open>\{)
    | (?P<close>\})
    | (?P<assign>(?P<lhs>[\w\*\.\->\(\)\[\]]+?)\s*=\s*(?P<rhs>[^;=]+);)
    """, re.X)


def walk(body):
    """yields (context, lhs, rhs) for every assignment; context = dict of the enclosing dim / FEType / switch cases.
    An `if` without braces governs the single statement that follows it."""
    text = re.sub(r"//[^\n]*", "", body)
    stack = []          # entries: dict(kind=..., depth=..., key=..., val=...)
    depth = 0
    pending = None      # an if/switch header waiting for its '{' (or, for if, a single statement)
    for m in TOKEN.finditer(text):
        if m.group("sw"):
            pending = dict(kind="switch", key=m.group("swvar"), val=None)
        elif m.group("if"):
            pending = dict(kind="if", key=m.group("cond").strip(), val=True)
        elif m.group("else"):
            pending = dict(kind="if", key="else", val=True)
        elif m.group("open"):
            depth += 1
            if pending is not None:
                pending["depth"] = depth
                stack.append(pending)
                pending = None
        elif m.group("close"):
            while stack and stack[-1].get("depth") == depth:
                stack.pop()
            depth -= 1
        elif m.group("case"):
            for e in reversed(stack):
                if e["kind"] == "switch":
                    e["val"] = int(m.group("caseval"))
                    break
        elif m.group("default"):
            for e in reversed(stack):
                if e["kind"] == "switch":
                    e["val"] = "default"
                    break
        elif m.group("assign"):
            ctx = {}
            for e in stack:
                if e["kind"] == "switch":
                    ctx[e["key"]] = e["val"]
                else:
                    ctx.setdefault("conds", []).append(e["key"])
            if pending is not None and pending["kind"] == "if":     # brace-less if: applies to this statement only
                ctx.setdefault("conds", []).append(pending["key"])
                ctx["guarded"] = pending["key"]
                pending = None
            yield ctx, m.group("lhs").strip(), m.group("rhs").strip()


def dim_of(conds):
    for c in conds or []:
        m = re.fullmatch(r"dim\s*==\s*(\d)", c)
        if m:
            return int(m.group(1))
    return None


def pyexpr(e):
    e = re.sub(r"p\s*(?:->|\.)\s*at\s*\(\s*(\d)\s*\)", r"p[\1]", e)
    e = re.sub(r"\bsqrt\b", "math.sqrt", e)
    e = re.sub(r"\bpow\b", "math.pow", e)
    return e


def quadrature(text):
    body = function_body(text, r"void\s+FE<SC,LO,GO,NO>::getQuadratureValues\s*\(")
    rules, remap = {}, {}
    for ctx, lhs, rhs in walk(body):
        conds = ctx.get("conds", [])
        dim = dim_of(conds)
        if dim not in (2, 3):
            continue
        if dim == 3 and not any("'P'" in c for c in conds):
            continue
        if lhs == "Degree" and "guarded" in ctx:         # if (Degree==3 || Degree==4) Degree=5;
            for d in re.findall(r"Degree\s*==\s*(\d+)", ctx["guarded"]):
                remap.setdefault(dim, {})[int(d)] = int(rhs)
            continue
        deg = ctx.get("Degree")
        if not isinstance(deg, int):
            continue
        r = rules.setdefault((dim, deg), dict(env={}, pts={}, w={}))
        m = re.fullmatch(r"QuadPts->at\((\d+)\)\.at\((\d+)\)", lhs)
        if m:
            r["pts"][(int(m.group(1)), int(m.group(2)))] = eval(pyexpr(rhs), {"math": math}, r["env"])
            continue
        m = re.fullmatch(r"QuadW->at\((\d+)\)", lhs)
        if m:
            r["w"][int(m.group(1))] = eval(pyexpr(rhs), {"math": math}, r["env"])
            continue
        if re.fullmatch(r"[A-Za-z]\w*", lhs):
            r["env"][lhs] = eval(pyexpr(rhs), {"math": math}, r["env"])
    out = {}
    for (dim, deg), r in sorted(rules.items()):
        nq = len(r["w"])
        assert nq > 0 and sorted(r["w"]) == list(range(nq)), (dim, deg)
        pts = [[r["pts"][(q, d)] for d in range(dim)] for q in range(nq)]
        out["%d,%d" % (dim, deg)] = dict(points=pts, weights=[r["w"][q] for q in range(nq)])
    return out, {str(d): {str(k): v for k, v in m.items()} for d, m in remap.items()}


def basis(text, quad):
    bphi = function_body(text, r"void\s+FE<SC,LO,GO,NO>::phi\s*\(")
    bgrad = function_body(text, r"void\s+FE<SC,LO,GO,NO>::gradPhi\s*\(")
    phi, grad = {}, {}
    for ctx, lhs, rhs in walk(bphi):
        dim = dim_of(ctx.get("conds"))
        if dim in (2, 3) and ctx.get("intFE") in (1, 2) and isinstance(ctx.get("i"), int) and lhs == "*value":
            phi[(dim, ctx["intFE"], ctx["i"])] = pyexpr(rhs)          # later assignments overwrite earlier ones, as in C++
    for ctx, lhs, rhs in walk(bgrad):
        dim = dim_of(ctx.get("conds"))
        m = re.fullmatch(r"value->at\((\d)\)", lhs)
        if dim in (2, 3) and ctx.get("intFE") in (1, 2) and isinstance(ctx.get("i"), int) and m:
            grad[(dim, ctx["intFE"], ctx["i"], int(m.group(1)))] = pyexpr(rhs)
    nen = {(2, 1): 3, (2, 2): 6, (3, 1): 4, (3, 2): 10}
    out = {}
    for (dim, fe), n in nen.items():
        # evaluation points: every quadrature point of this dimension plus a few generic interior points
        pts = [p for key, r in quad.items() if key.startswith("%d," % dim) for p in r["points"]]
        pts += [[0.1, 0.2, 0.3][:dim], [0.25, 0.5, 0.125][:dim], [1.0 / 3.0] * dim]
        vals, grads = [], []
        for p in pts:
            env = {"p": list(p) + [0.0] * (3 - dim), "math": math}
            vals.append([eval(phi[(dim, fe, i)], env) for i in range(n)])
            grads.append([[eval(grad[(dim, fe, i, d)], env) for d in range(dim)] for i in range(n)])
        out["%d,P%d" % (dim, fe)] = dict(points=pts, phi=vals, grad=grads)
    return out


def element_tables(text):
    """P1 element tables of the structured generators as corner offsets.  buildMesh2D / buildMesh3D write, per cell
    (s, r[, t]) and per element, statements `(*elementsVec)[counter][k] = <expr in r, s, t, M>`-like; instead of
    interpreting the index algebra here, the cell's node ids are evaluated for one interior cell of a 3x3(x3) block
    and converted to corner offsets (dr, ds[, dt])."""
    out = {}
    for fn, dim in (("buildMesh2D", 2), ("buildMesh3D", 3)):
        body = function_body(text, r"void\s+MeshStructured<SC,LO,GO,NO>::%s\s*\(" % fn)
        nen, per_cell, M = dim + 1, (2 if dim == 2 else 6), 3
        P = M + 1
        env = dict(M=M, r=1, s=1, t=1)
        tok = re.compile(r"\(\*elementsVec\)\[\s*counter\s*\]\[\s*(\d)\s*\]\s*=\s*([^;]+);|(counter\s*\+\+)")
        elems, cur, started = [], {}, False
        for m in tok.finditer(body):
            if m.group(3):                       # counter++ closes the current element
                if started and cur:
                    elems.append([cur[k] for k in range(nen)])
                    cur = {}
                    if len(elems) == per_cell:
                        break
                continue
            started = True
            nid = int(eval(m.group(2), {}, dict(env)))
            if dim == 2:
                cur[int(m.group(1))] = [nid % P - env["r"], nid // P - env["s"]]
            else:
                cur[int(m.group(1))] = [nid % P - env["r"], (nid // P) % P - env["s"], nid // (P * P) - env["t"]]
        assert len(elems) == per_cell, (fn, len(elems))
        out[str(dim)] = dict(cell_elements=elems)
    return out


def main():
    fe = open(FE_DEF).read()
    quad, remap = quadrature(fe)
    tables = dict(source="generated by tests/golden/make_ref_tables.py from the numeric literals of "
                         "feddlib/core/FE/FE_def.hpp (getQuadratureValues, phi, gradPhi) and "
                         "feddlib/core/Mesh/MeshStructured_def.hpp (buildMesh2D/3D); numbers only",
                  quadrature=quad, degree_remap=remap, basis=basis(fe, quad))
    try:
        tables["structured_cells"] = element_tables(open(MESH_DEF).read())
    except Exception as e:          # the element tables are a bonus: never lose the FE tables over them
        tables["structured_cells"] = {"error": repr(e)}
    json.dump(tables, open(OUT, "w"), indent=0)
    print("wrote", OUT)
    for k, r in quad.items():
        print("rule", k, "nq", len(r["weights"]), "sum w %.15g" % sum(r["weights"]))
    print("remap", remap)
    print("cells", tables["structured_cells"])


if __name__ == "__main__":
    sys.exit(main())
