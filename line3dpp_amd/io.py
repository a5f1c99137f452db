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


# ---- BIN result format (Line3D::save3DLinesAsBIN, line3D.cc:2690-2711) ------------------------------------------
# boost::archive::binary_oarchive of std::vector<FinalLine3D> (serialization.h:38-45), little endian, as written by
# the Boost version behind the reference's fixtures (archive library version 10).  Layout, verified byte for byte
# against testdata/Line3D++_ref/*vis_3.bin (tests/test_bin_format.py):
#   u64 22, "serialization::archive", u16 library version, sizeof(int, long, float, double) as 4 bytes, u32 1 (endian)
#   every class writes a 5-byte header (u8 tracking = 0, u32 version = 0) at its FIRST occurrence in the archive only
#   collections: u64 count, u32 item_version
#   vector<FinalLine3D>   = [hdr] count item_version FinalLine3D*
#   FinalLine3D           = [hdr] list<Segment3D> LineCluster3D                       (segment3D.h:165-178)
#   list<Segment3D>       = [hdr] count item_version Segment3D*
#   Segment3D             = [hdr] f32 length_, u8 valid_, 9 x f64 (P1, P2, dir)       (segment3D.h:99-115)
#   LineCluster3D         = [hdr] Segment3D list<Segment2D> u32 reference_view_        (segment3D.h:152-160)
#   list<Segment2D>       = [hdr] count item_version Segment2D*
#   Segment2D             = [hdr] u32 camID_, u32 segID_                               (commons.h:123-130)
_BIN_SIG = b"serialization::archive"
_CLASS_HDR = b"\x00" * 5


class _Reader:
    def __init__(self, buf):
        self.b, self.p, self.seen = buf, 0, set()

    def take(self, fmt):
        import struct
        try:
            v = struct.unpack_from("<" + fmt, self.b, self.p)
        except struct.error as e:
            raise ValueError(f"truncated archive at byte {self.p}: {e}") from None
        self.p += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def hdr(self, cls):
        if cls not in self.seen:
            self.seen.add(cls)
            if self.b[self.p:self.p + 5] != _CLASS_HDR:
                raise ValueError(f"unexpected class header for {cls} at byte {self.p}")
            self.p += 5

    def seg3d(self):
        self.hdr("Segment3D")
        length, valid = self.take("fB")
        geo = np.array(self.take("9d"))
        return length, valid, geo


def read_3d_lines_bin(path):
    """-> (lines, library_version); lines = list of dicts: segments [n,9] float64 (P1, P2, dir), seg_length [n] float32,
    seg_valid [n] uint8, cluster_line [9] float64, cluster_length, cluster_valid, residuals [m,2] uint32 (camID, segID),
    reference_view"""
    r = _Reader(open(path, "rb").read())
    if r.take("Q") != len(_BIN_SIG) or r.b[r.p:r.p + len(_BIN_SIG)] != _BIN_SIG:
        raise ValueError(f"{path}: not a boost binary archive")
    r.p += len(_BIN_SIG)
    lib_version = r.take("H")
    if bytes(r.b[r.p:r.p + 8]) != bytes([4, 8, 4, 8, 1, 0, 0, 0]):
        raise ValueError(f"{path}: written on a platform with other type sizes / endianness")
    r.p += 8
    r.hdr("vector<FinalLine3D>")
    n, _ = r.take("QI")
    lines = []
    for _ in range(n):
        r.hdr("FinalLine3D")
        r.hdr("list<Segment3D>")
        ns, _ = r.take("QI")
        segs = [r.seg3d() for _ in range(ns)]
        r.hdr("LineCluster3D")
        cl_len, cl_valid, cl_geo = r.seg3d()
        r.hdr("list<Segment2D>")
        nr, _ = r.take("QI")
        res = np.zeros((nr, 2), np.uint32)
        for k in range(nr):
            r.hdr("Segment2D")
            res[k] = r.take("II")
        ref_view = r.take("I")
        lines.append(dict(segments=np.array([s[2] for s in segs]).reshape(-1, 9),
                          seg_length=np.array([s[0] for s in segs], np.float32),
                          seg_valid=np.array([s[1] for s in segs], np.uint8),
                          cluster_line=cl_geo, cluster_length=np.float32(cl_len), cluster_valid=int(cl_valid),
                          residuals=res, reference_view=int(ref_view)))
    if r.p != len(r.b):
        raise ValueError(f"{path}: {len(r.b) - r.p} trailing bytes")
    return lines, lib_version


def format_3d_lines_bin(lines, lib_version=10):
    """inverse of read_3d_lines_bin: the exact bytes Line3D::save3DLinesAsBIN writes for these lines"""
    import struct
    out = [struct.pack("<Q", len(_BIN_SIG)), _BIN_SIG, struct.pack("<H", lib_version), bytes([4, 8, 4, 8, 1, 0, 0, 0])]
    seen = set()

    def hdr(cls):
        if cls not in seen:
            seen.add(cls)
            out.append(_CLASS_HDR)

    def seg3d(length, valid, geo):
        hdr("Segment3D")
        out.append(struct.pack("<fB9d", float(length), int(valid), *[float(x) for x in geo]))
    hdr("vector<FinalLine3D>")
    out.append(struct.pack("<QI", len(lines), 0))
    for L in lines:
        hdr("FinalLine3D")
        hdr("list<Segment3D>")
        out.append(struct.pack("<QI", len(L["segments"]), 0))
        for g, ln, va in zip(L["segments"], L["seg_length"], L["seg_valid"]):
            seg3d(ln, va, g)
        hdr("LineCluster3D")
        seg3d(L["cluster_length"], L["cluster_valid"], L["cluster_line"])
        hdr("list<Segment2D>")
        out.append(struct.pack("<QI", len(L["residuals"]), 0))
        for cam, seg in L["residuals"]:
            hdr("Segment2D")
            out.append(struct.pack("<II", int(cam), int(seg)))
        out.append(struct.pack("<I", int(L["reference_view"])))
    return b"".join(out)


# ---- input side: the segment cache and VisualSfM .nvm files ---------------------------------------
# Segment cache = what Line3D::detectLineSegments stores / loads per image when `load_segments` is set
# (line3D.cc:295-309, 362-366): "<data folder>/segments_L3D++_<camID>_<width>x<height>_<max segments>.bin", a
# boost::archive::binary_oarchive of L3DPP::DataArray<float4> (dataArray.h:352-374): width_, height_, real_width_
# (u32), pitchCPU_, strideCPU_, pitchGPU_, strideGPU_ (u64), then real_width_ * height_ float4 elements, each through
# serialize(float4) (dataArray.h:62-69: the class header of float4 appears once, before the first element).
# width_ = number of segments, height_ = 1; the host row is padded to a multiple of 32 bytes (dataArray.h:111-122), so
# an odd number of segments carries one padding element.
def segment_cache_name(camID, width, height, max_segments=3000):
    return f"segments_L3D++_{camID}_{width}x{height}_{max_segments}.bin"


def _data_array_geometry(n):
    pitch = n * 16
    real = n + (0 if pitch % 32 == 0 else (32 - pitch % 32) // 16)
    return real, real * 16, real           # real_width_, pitchCPU_, strideCPU_


def format_segment_cache(segs, lib_version=10):
    """the bytes serializeToFile(name, DataArray<float4>(n, 1, false, segments)) writes for [n,4] float32 (x1,y1,x2,y2)"""
    import struct
    segs = np.ascontiguousarray(segs, np.float32).reshape(-1, 4)
    n = len(segs)
    real, pitch, stride = _data_array_geometry(n)
    data = np.zeros((real, 4), np.float32)
    data[:n] = segs
    return b"".join([struct.pack("<Q", len(_BIN_SIG)), _BIN_SIG, struct.pack("<H", lib_version), bytes([4, 8, 4, 8, 1, 0, 0, 0]),
                     _CLASS_HDR, struct.pack("<IIIQQQQ", n, 1, real, pitch, stride, 0, 0),
                     _CLASS_HDR if real else b"", data.tobytes()])


def read_segment_cache(path):
    """-> [n,4] float32 segments of a cache file written by the reference (or by format_segment_cache)"""
    r = _Reader(open(path, "rb").read())
    if r.take("Q") != len(_BIN_SIG) or r.b[r.p:r.p + len(_BIN_SIG)] != _BIN_SIG:
        raise ValueError(f"{path}: not a boost binary archive")
    r.p += len(_BIN_SIG)
    r.take("H")
    if bytes(r.b[r.p:r.p + 8]) != bytes([4, 8, 4, 8, 1, 0, 0, 0]):
        raise ValueError(f"{path}: written on a platform with other type sizes / endianness")
    r.p += 8
    r.hdr("DataArray<float4>")
    width, height, real, pitch, stride, _, _ = r.take("IIIQQQQ")
    if height != 1 or real < width or pitch != real * 16 or stride != real:
        raise ValueError(f"{path}: not a one-row DataArray<float4> (width {width}, height {height}, real width {real})")
    if real:
        r.hdr("float4")
    if r.p + real * 16 > len(r.b):
        raise ValueError(f"{path}: truncated ({len(r.b) - r.p} of {real * 16} element bytes)")
    data = np.frombuffer(r.b, np.float32, real * 4, r.p).reshape(real, 4)
    r.p += real * 16
    if r.p != len(r.b):
        raise ValueError(f"{path}: {len(r.b) - r.p} trailing bytes")
    return data[:width].copy()


# VisualSfM .nvm as main_vsfm.cpp:144-250 reads it: two ignored lines, the number of cameras, one line per camera
# (file name, focal length, quaternion w x y z, camera centre, radial distortion, 0), an ignored line, the number of 3D
# points, one line per point (position, colour, number of measurements, then per measurement camera index, feature
# index, x, y).
def nvm_rotation(qw, qx, qy, qz):
    """main_vsfm.cpp:188-199"""
    return np.array([[1.0 - 2.0 * qy * qy - 2.0 * qz * qz, 2.0 * qx * qy - 2.0 * qz * qw, 2.0 * qx * qz + 2.0 * qy * qw],
                     [2.0 * qx * qy + 2.0 * qz * qw, 1.0 - 2.0 * qx * qx - 2.0 * qz * qz, 2.0 * qy * qz - 2.0 * qx * qw],
                     [2.0 * qx * qz - 2.0 * qy * qw, 2.0 * qy * qz + 2.0 * qx * qw, 1.0 - 2.0 * qx * qx - 2.0 * qy * qy]])


def _mv3(M, v):
    """M v for a 3x3 M in the evaluation order of the reference's fixed-size Eigen product (and of the C-ABI readers):
    (M[i,0] v[0] + M[i,1] v[1]) + M[i,2] v[2] -- numpy's matmul may sum in another order or fuse the multiplies"""
    M = np.asarray(M, np.float64); v = np.asarray(v, np.float64)
    return (M[:, 0] * v[0] + M[:, 1] * v[1]) + M[:, 2] * v[2]


def read_nvm(path):
    """-> list of cameras in file order (the reference uses the index as camID): dict(filename, focal, R, t, C,
    distortion, worldpoints = ids of the 3D points it sees, median_depth = sorted distances to them [n/2] as float32,
    main_vsfm.cpp:300-303; None for a camera without points, which the reference skips)"""
    with open(path) as f:
        lines = f.read().split("\n")
    pos = 2
    n_cams = int(lines[pos].split()[0]); pos += 1
    if n_cams == 0:
        raise ValueError("No aligned cameras in NVM file!")          # main_vsfm.cpp:157-161
    cams = []
    for i in range(n_cams):
        tok = lines[pos].split(); pos += 1
        focal, qw, qx, qy, qz, cx, cy, cz, dist = (float(x) for x in tok[1:10])
        R = nvm_rotation(qw, qx, qy, qz)
        Cc = np.array([cx, cy, cz])
        cams.append(dict(filename=tok[0], focal=np.float32(focal), R=R, t=_mv3(-R, Cc), C=Cc, distortion=np.float32(dist),   # t = -R*C, :207
                         worldpoints=[], _depths=[]))
    pos += 1
    n_pts = int(lines[pos].split()[0]); pos += 1
    for i in range(n_pts):
        if pos >= len(lines):
            break                      # the file ends early: the stream parser of main_vsfm.cpp sees no further measurements
        tok = lines[pos].split(); pos += 1
        if len(tok) < 7:
            continue
        p = np.array([float(tok[0]), float(tok[1]), float(tok[2])])
        nv = int(tok[6])
        for j in range(nv):
            cam = int(tok[7 + 4 * j])
            if cam >= n_cams:
                raise ValueError("malformed measurement in NVM file")
            cams[cam]["worldpoints"].append(i)
            cams[cam]["_depths"].append(np.float32(np.linalg.norm(p - cams[cam]["C"])))
    for c in cams:
        d = sorted(c.pop("_depths"))
        c["median_depth"] = d[len(d) // 2] if d else None
    return cams


def nvm_intrinsics(focal, width, height):
    """K as main_vsfm.cpp:272-282 builds it: principal point at the image centre (float arithmetic there)"""
    return np.array([[np.float32(focal), 0.0, np.float32(width) / np.float32(2.0)],
                     [0.0, np.float32(focal), np.float32(height) / np.float32(2.0)], [0.0, 0.0, 1.0]], np.float64)


# ---- COLMAP text results and bundler files (the Python twin of l3d_sfm_open_colmap / l3d_sfm_open_bundler) --------------
def rotation_from_q(qw, qx, qy, qz):
    """Line3D::rotationFromQ, line3D.cc:2730-2754"""
    n = qw * qw + qx * qx + qy * qy + qz * qz
    s = 0.0 if abs(n) < 1e-12 else 2.0 / n
    wx, wy, wz = s * qw * qx, s * qw * qy, s * qw * qz
    xx, xy, xz = s * qx * qx, s * qx * qy, s * qx * qz
    yy, yz, zz = s * qy * qy, s * qy * qz, s * qz * qz
    return np.array([[1.0 - (yy + zz), xy - wz, xz + wy], [xy + wz, 1.0 - (xx + zz), yz - wx], [xz - wy, yz + wx, 1.0 - (xx + yy)]])


_COLMAP_MODELS = {  # parameter order of cameras.txt -> (fx, fy, cx, cy, k1, k2, p1, p2, k3), main_colmap.cpp:177-219
    "SIMPLE_PINHOLE": lambda p: (p[0], p[0], p[1], p[2], 0, 0, 0, 0, 0),
    "PINHOLE": lambda p: (p[0], p[1], p[2], p[3], 0, 0, 0, 0, 0),
    "SIMPLE_RADIAL": lambda p: (p[0], p[0], p[1], p[2], p[3], 0, 0, 0, 0),
    "RADIAL": lambda p: (p[0], p[0], p[1], p[2], p[3], p[4], 0, 0, 0),
    "OPENCV": lambda p: (p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], 0),
    "FULL_OPENCV": lambda p: (p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8]),
}


def _median_depth(C, pts):
    d = sorted(np.float32(np.linalg.norm(C - p)) for p in pts)
    return d[len(d) // 2] if d else None


def read_colmap(folder):
    """cameras.txt / images.txt / points3D.txt as main_colmap.cpp:136-348 reads them -> list of images in file order:
    dict(id, camera, name, width, height, K, R, t, C, radial (k1, k2, k3), tangential (p1, p2), worldpoints,
    median_depth or None).  An image whose camera is unknown is dropped; points3D.txt lines that do not parse as
    "id X Y Z" are ignored; a worldpoint without an entry there sits at the origin (the reference's map default)."""
    import os
    def getlines(path):      # std::getline's view of a file: no extra empty line behind a final newline
        text = open(path).read()
        lines = text.split("\n")
        return lines[:-1] if lines and lines[-1] == "" else lines
    cams = {}
    for line in getlines(os.path.join(folder, "cameras.txt")):
        if line[:1] == "#":
            continue
        tok = line.split()
        # (a blank line is NOT skipped: the reference and l3d_sfm_open_colmap parse it and fail on its empty model name)
        model = tok[1] if len(tok) > 1 else ""
        if model not in _COLMAP_MODELS:
            raise ValueError(f"camera model {model} unknown!")
        fx, fy, cx, cy, k1, k2, p1, p2, k3 = (float(x) for x in _COLMAP_MODELS[tok[1]]([float(x) for x in tok[4:]] + [0.0] * 9))
        cams[int(tok[0])] = dict(width=int(tok[2]), height=int(tok[3]), K=np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]]),
                                 radial=np.array([k1, k2, k3]), tangential=np.array([p1, p2]))
    imgs, by_id, wps = [], {}, {}
    first, cur = True, None
    for line in open(os.path.join(folder, "images.txt")).read().split("\n"):
        if line[:1] == "#":
            continue
        tok = line.split()
        if first:
            cur = None
            if len(tok) >= 9 and int(tok[8]) in cams:
                cam = cams[int(tok[8])]
                R = rotation_from_q(*(float(x) for x in tok[1:5]))
                t = np.array([float(x) for x in tok[5:8]])
                cur = dict(id=int(tok[0]), camera=int(tok[8]), name=tok[9] if len(tok) > 9 else "", R=R, t=t, C=R.T @ (-1.0 * t),
                           worldpoints=[], **cam)
                by_id[cur["id"]] = cur
                imgs.append(cur)
            first = False
        else:
            if cur is not None:
                lst = []
                for k in range(2, len(tok), 3):
                    wp = int(tok[k])
                    if wp >= 0:
                        lst.append(wp); wps[wp] = np.zeros(3)
                cur["worldpoints"] = lst
            first = True
    # a repeated IMAGE_ID: the reference's maps are keyed by the id, so BOTH entries of the image sequence see the last pose
    # and the last worldpoint list (l3d_sfm_open_colmap does the same fix-up)
    imgs = [im if by_id[im["id"]] is im else dict(by_id[im["id"]]) for im in imgs]
    for line in open(os.path.join(folder, "points3D.txt")).read().split("\n"):
        tok = line.split()
        try:
            pid, X, Y, Z = int(tok[0]), float(tok[1]), float(tok[2]), float(tok[3])
        except (ValueError, IndexError):
            continue
        if pid in wps:
            wps[pid] = np.array([X, Y, Z])
    for im in imgs:
        im["median_depth"] = _median_depth(im["C"], [wps[w] for w in im["worldpoints"]])
    return imgs


def read_bundler(path):
    """bundle.rd.out as main_bundler.cpp:147-252 reads it -> list of cameras (index = camID): dict(id, focal, radial
    (d1, d2, 0), R, t (second and third row / entry negated), C, worldpoints, median_depth or None)"""
    lines = open(path).read().split("\n")
    n_cams, n_pts = (int(x) for x in lines[1].split()[:2])
    if n_cams == 0 or n_pts == 0:
        raise ValueError("No cameras and/or points in bundle file!")
    pos, cams = 2, []
    for i in range(n_cams):
        f, d1, d2 = (float(x) for x in lines[pos].split()[:3])
        R = np.array([[float(x) for x in lines[pos + 1 + j].split()[:3]] for j in range(3)])
        R[1] *= -1.0; R[2] *= -1.0
        t = np.array([float(x) for x in lines[pos + 4].split()[:3]])
        t[1] *= -1.0; t[2] *= -1.0
        cams.append(dict(id=i, focal=np.float32(f), radial=np.array([np.float32(d1), np.float32(d2), 0.0]), R=R, t=t,
                         C=R.T @ (-1.0 * t), worldpoints=[], _pts=[]))
        pos += 5
    for i in range(n_pts):
        if pos + 2 >= len(lines):
            break
        p = np.array([float(x) for x in lines[pos].split()[:3]])
        tok = lines[pos + 2].split()
        for j in range(int(tok[0])):
            cam = int(tok[1 + 4 * j])
            if cam >= n_cams:
                raise ValueError("malformed view list in bundle file")
            cams[cam]["worldpoints"].append(i); cams[cam]["_pts"].append(p)
        pos += 3
    for c in cams:
        c["median_depth"] = _median_depth(c["C"], c.pop("_pts"))
    return cams
