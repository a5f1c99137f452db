"""Reader/formatter of the reference's TXT result format (Line3D::save3DLinesAsTXT, line3D.cc:2631-2688):
one text line per 3D line --  n_segments (P1.x P1.y P1.z P2.x P2.y P2.z)*  n_residuals (camID segID x1 y1 x2 y2)*
with the C++ stream defaults (6 significant digits).  The library writes the format itself
(l3d_save_3d_lines_txt); this module parses it (e.g. testdata/Line3D++_ref/*.txt of the reference) and
re-creates the text for diffing."""
import numpy as np


def read_3d_lines_txt(path):
    """-> list of dicts: segments [n,6] float64 (P1, P2), residuals [m,2] uint32 (camID, segID), coords2D [m,4] float32"""
    out = []
    with open(path) as f:
        for raw in f:
            tok = raw.split()
            if not tok:
                continue
            n = int(tok[0]); p = 1
            segs = np.array(tok[p:p + 6 * n], np.float64).reshape(n, 6); p += 6 * n
            m = int(tok[p]); p += 1
            rec = np.array(tok[p:p + 6 * m], np.float64).reshape(m, 6); p += 6 * m
            if p != len(tok):
                raise ValueError(f"{path}: trailing tokens in a 3D-line record")
            out.append(dict(segments=segs, residuals=rec[:, :2].astype(np.uint32), coords2D=rec[:, 2:].astype(np.float32)))
    return out


def _g(x):
    return "%g" % x          # == operator<<(std::ostream&, double/float) with the default precision of 6


def format_3d_lines_txt(lines):
    """inverse of read_3d_lines_txt: the exact text Line3D::save3DLinesAsTXT writes for these lines"""
    rows = []
    for L in lines:
        if len(L["segments"]) == 0:
            continue
        t = [str(len(L["segments"]))]
        for s in L["segments"]:
            t += [_g(v) for v in s]
        t.append(str(len(L["residuals"])))
        for (cam, seg), co in zip(L["residuals"], L["coords2D"]):
            t += [str(int(cam)), str(int(seg))] + [_g(np.float32(v)) for v in co]
        rows.append(" ".join(t) + " \n")
    return "".join(rows)


def format_obj(lines):
    """the text Line3D::saveResultAsOBJ (line3D.cc:2579-2628) writes: two `v` records per 3D segment, then one `l`
    record per segment"""
    v, n = [], 0
    for L in lines:
        for s in L["segments"]:
            v.append("v " + " ".join(_g(x) for x in s[:3]) + "\n")
            v.append("v " + " ".join(_g(x) for x in s[3:]) + "\n")
            n += 1
    return "".join(v) + "".join(f"l {2 * k + 1} {2 * k + 2}\n" for k in range(n))


def format_stl(lines):
    """the text Line3D::saveResultAsSTL (line3D.cc:2465-2531) writes (degenerate triangles P1-P2-P1, printf %e)"""
    t = ["solid lineModel\n"]
    for L in lines:
        for s in L["segments"]:
            a = ["%e" % x for x in s]
            t += [" facet normal 1.0e+000 0.0e+000 0.0e+000\n", "  outer loop\n",
                  "   vertex %s %s %s\n" % tuple(a[:3]), "   vertex %s %s %s\n" % tuple(a[3:]),
                  "   vertex %s %s %s\n" % tuple(a[:3]), "  endloop\n", " endfacet\n"]
    t.append("endsolid lineModel\n")
    return "".join(t)
