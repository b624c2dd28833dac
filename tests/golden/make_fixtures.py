#!/usr/bin/env python3
"""Extract the reference's own known-answer data into small JSON fixtures.

Runs ONLY in the build container (needs /root/reference).  It reads the
*data* files the reference's tests hold -- literal numeric tables, no code --
and writes them as JSON under tests/golden/.  The GPU box never runs this.

Sources (all under /root/reference/test/):
  vlidort_baseline/reference_data/siewert2000_IIA_greek.jl     Greek coefficients, Siewert 2000 problem IIA
  vlidort_baseline/reference_data/siewert2000_IIA_truth.jl     VLIDORT 2.8.3 tables 2-9
  vlidort_baseline/reference_data/solar_tester_atmosphere.jl   23-layer atmosphere
  vlidort_baseline/reference_data/solar_tester_truth.jl        scalar truth  [geom, level, dir, task]
  vlidort_baseline/reference_data/solar_tester_problemIII_aerosol.jl
  vlidort_baseline/reference_data/solar_tester_vector_truth.jl IQU truth
  benchmarks/natraj_trues.jl                                   Natraj 2009 Rayleigh tau=0.5 I,Q,U
  benchmarks/6SV1_R_trues.jl                                   6SV1 reflectances, 6 cases
  reference/phase1b_RRS_sanghavi_q0.jld2                       rotational-Raman regression arrays R, T, ieR, ieT
  test_parameters/Phase1b_RRS_761-764nm.yaml                   the scene of that regression (profile, geometry)
and one data table from src/ that the Cox-Munk oracle needs:
  src/CoreRT/Surfaces/water_refraction.jl                      Segelstein (1981) water refractive index table
and the parsed parameters of one shipped scene file:
  config/ocean_coxmunk.yaml                                    BASELINE config C3 (Cox-Munk ocean, 33 layers, IQUV)

The procedure each fixture is used with (geometry, tolerances) is recorded in
the fixture's "procedure" field with the reference file:line it restates.
"""
import json
import os
import re
import sys

import numpy as np

REF = os.environ.get("VSM_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _strip_comments(text):
    return re.sub(r"#[^\n]*", "", text)


def _numbers(s):
    return [float(x) for x in re.findall(r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?", s)]


def _block(text, name):
    """Return the bracketed literal following `name =` (balanced brackets)."""
    m = re.search(r"\b" + re.escape(name) + r"\s*=\s*", text)
    if not m:
        raise KeyError(name)
    i = text.index("[", m.end())
    depth, j = 0, i
    while True:
        c = text[j]
        if c == "[":
            depth += 1
        elif c == "]":
            depth -= 1
            if depth == 0:
                break
        j += 1
    return text[i + 1:j]


def _matrix(text, name):
    """Julia matrix literal: rows split by ';' (or newlines), entries by blanks."""
    body = _block(text, name)
    rows = [r for r in re.split(r"[;\n]", body) if r.strip()]
    return [_numbers(r) for r in rows]


def _vector(text, name):
    return _numbers(_block(text, name))


def _read(rel):
    with open(os.path.join(REF, rel), encoding="utf-8") as f:
        return _strip_comments(f.read())


def siewert():
    g = _read("test/vlidort_baseline/reference_data/siewert2000_IIA_greek.jl")
    rows = {k: _numbers(re.search(r"row%d\s*=\s*\[(.*?)\]" % k, g, re.S).group(1)) for k in range(1, 7)}
    t = _read("test/vlidort_baseline/reference_data/siewert2000_IIA_truth.jl")
    tables = {str(k): _matrix(t, "SIEWERT_TABLE%d" % k) for k in range(2, 10)}
    cos = _vector(t, "SIEWERT_TABLE2_COSINES")
    fx = {
        "source": "test/vlidort_baseline/reference_data/siewert2000_IIA_{greek,truth}.jl",
        "procedure": {
            "ref": "test/vlidort_baseline/cases/case_A_siewert2000.jl:29-123, configs/siewert2000_IIA.yaml",
            "tau_aer": 1.0, "tau_rayl": 0.0, "ssa": 0.973527, "f_trunc": 0.0,
            "sza_deg": 53.130102354065, "albedo": 0.0, "l_trunc": 30, "max_m": 30,
            "polarization": "IQUV",
            "vza_deg": [0.0001, 25.841932763, 36.869897646, 45.572995999, 53.130102354, 60.0,
                        66.421821522, 72.542396876, 78.463040967, 84.260829523, 89.9999],
            "azimuths_deg": [0.0, 90.0, 180.0],
            "compare": "pi*R[:,stokes,0] vs table[row(-|mu|), col 0]; truth Q,U,V sign-flipped",
            "rtol": {"I": 5e-4, "Q": 1e-2, "U": 5e-4, "V": 5e-4},
            "atol_rule": "100*eps(FT)*max|truth| added to the denominator",
        },
        # mapping per siewert2000_IIA_greek.jl header: alpha=row1, beta=row2, gamma=-row3,
        # delta=row4, epsilon=-row5, zeta=row6
        "greek": {
            "alpha": rows[1], "beta": rows[2], "gamma": [-x for x in rows[3]],
            "delta": rows[4], "epsilon": [-x for x in rows[5]], "zeta": rows[6],
        },
        "table_cosines": cos,
        "tau_levels": _vector(t, "SIEWERT_TAU_LEVELS"),
        # (azimuth, stokes) -> table number, from siewert2000_IIA_truth.jl header
        "table_of": {"0.0:I": 2, "0.0:Q": 3, "90.0:I": 4, "90.0:Q": 5, "90.0:U": 6, "90.0:V": 7,
                     "180.0:I": 8, "180.0:Q": 9},
        "tables": tables,
    }
    return "siewert2000_IIA.json", fx


def natraj():
    t = _read("test/benchmarks/natraj_trues.jl")
    fx = {
        "source": "test/benchmarks/natraj_trues.jl",
        "procedure": {
            "ref": "test/test_CoreRT.jl:110-157, test/benchmarks/natraj.yaml",
            "tau_rayl": 0.5, "depol": 0.0, "albedo": 0.0, "nstreams": 11, "polarization": "IQUV",
            "mu0": 0.2,
            "mu_view": [0.02, 0.06, 0.10, 0.16, 0.20, 0.28, 0.32, 0.40, 0.52, 0.64, 0.72, 0.84, 0.92,
                        0.96, 0.98, 1.00],
            "azimuths_deg": [0.0, 30.0, 60.0, 90.0, 120.0, 150.0, 180.0],
            "compare": "pi*R vs trues[mu, az]",
            "rtol": {"I": 5e-4, "Q": 2.5e-3, "U": 5e-4},
            "mask": "Q,U compared only where modeled >= 0.01",
        },
        "I": _matrix(t, "I_trues"), "Q": _matrix(t, "Q_trues"), "U": _matrix(t, "U_trues"),
    }
    return "natraj2009.json", fx


def sixsv():
    t = _read("test/benchmarks/6SV1_R_trues.jl")
    body = "[" + _block(t, "R_trues") + "]"
    fx = {
        "source": "test/benchmarks/6SV1_R_trues.jl",
        "procedure": {
            "ref": "test/test_CoreRT.jl:7-43, test/benchmarks/6SV1_1.yaml",
            "depol": 0.0, "nstreams": 11, "polarization": "IQUV",
            "vza_deg": [0.0, 11.4783, 16.2602, 23.0739, 32.8599, 43.9455, 50.2082, 58.6677, 66.4218,
                        71.3371, 73.7398, 78.463, 80.7931, 84.2608, 86.5602, 88.854],
            "azimuths_deg": [180.0, 90.0, 0.0],
            "cases": [
                {"sza_deg": [23.0739, 53.1301, 78.4630], "tau": 0.1, "albedo": 0.0},
                {"sza_deg": [0.0001, 36.8699, 66.4218], "tau": 0.1, "albedo": 0.25},
                {"sza_deg": [0.0001, 36.8699, 66.4218], "tau": 0.25, "albedo": 0.0},
                {"sza_deg": [23.0739, 53.1301, 78.4630], "tau": 0.25, "albedo": 0.25},
                {"sza_deg": [23.0739, 53.1301, 78.4630], "tau": 0.50, "albedo": 0.0},
                {"sza_deg": [0.0001, 36.8699, 66.4218], "tau": 0.50, "albedo": 0.25},
            ],
            "compare": "pi*R[:,0,0]/mu0 vs R_trues[case][sza][az][vza]",
            "rtol": 6e-3,
        },
        "R_trues": json.loads(body),
    }
    return "sixsv1.json", fx


def _solar_atmos():
    a = _read("test/vlidort_baseline/reference_data/solar_tester_atmosphere.jl")
    return {
        "height_km": _vector(a, "SOLAR_TESTER_HEIGHT_KM"),
        "molext": _vector(a, "SOLAR_TESTER_MOLEXT"),
        "molomg": _vector(a, "SOLAR_TESTER_MOLOMG"),
    }


def _reshape4(vals):
    arr = np.asarray(vals, dtype=np.float64).reshape((36, 5, 2, 6), order="F")
    # keep task 1 (pp_noDM), TOA-up (level 1, dir 1) and BOA-down (level 5, dir 2)
    return {"toa_up": arr[:, 0, 0, 0].tolist(), "boa_dn": arr[:, 4, 1, 0].tolist()}


def solar_scalar():
    t = _read("test/vlidort_baseline/reference_data/solar_tester_truth.jl")
    fx = {
        "source": "test/vlidort_baseline/reference_data/solar_tester_{atmosphere,truth}.jl",
        "procedure": {
            "ref": "test/vlidort_baseline/cases/case_B_solar_tester.jl:19-161, configs/solar_tester.yaml",
            "polarization": "I", "l_trunc": 15, "max_m": 16, "depol": 0.01072, "albedo": 0.05,
            "aerosol": {"g": 0.8, "ssa": 0.95, "tau_total": 0.5, "nmoments": 15, "layers": "18..23"},
            "sza_deg": [35.0, 67.0, 75.0, 82.0], "vza_deg": [10.0, 20.0, 40.0], "raz_deg": [0.0, 90.0, 180.0],
            "geom_index": "(i_sza)*9 + (i_vza)*3 + i_raz  (0-based)",
            "compare": "R[vza,0,0] vs toa_up[geom]; T[vza,0,0] vs boa_dn[geom]  (no pi)",
            "gated_geometry": {"sza_deg": 35.0, "raz_deg": 0.0},
            "rtol": 1e-3,
        },
        "atmosphere": _solar_atmos(),
        "truth": _reshape4(_vector(t, "SOLAR_TESTER_STOKES")),
    }
    return "solar_tester_scalar.json", fx


def solar_vector():
    t = _read("test/vlidort_baseline/reference_data/solar_tester_vector_truth.jl")
    p = _read("test/vlidort_baseline/reference_data/solar_tester_problemIII_aerosol.jl")
    n = 16
    a1, b1, a2, a3, b2, a4 = (_vector(p, "PROBLEMIII_" + k)[:n] for k in ("a1", "b1", "a2", "a3", "b2", "a4"))
    fx = {
        "source": "test/vlidort_baseline/reference_data/solar_tester_{atmosphere,problemIII_aerosol,vector_truth}.jl",
        "procedure": {
            "ref": "test/vlidort_baseline/cases/case_C_solar_tester_vector.jl:19-173, configs/solar_tester_vector.yaml",
            "polarization": "IQU", "l_trunc": 15, "max_m": 16, "depol": 0.01072, "albedo": 0.05,
            "aerosol": {"ssa": 0.99999, "tau_total": 0.5, "nmoments": 15, "layers": "18..23"},
            "sza_deg": [35.0, 67.0, 75.0, 82.0], "vza_deg": [10.0, 20.0, 40.0], "raz_deg": [10.0, 90.0, 170.0],
            "compare": "R[vza,s,0] vs toa_up; T[vza,s,0] vs boa_dn; truth Q,U sign-flipped",
            "gated_geometry": {"sza_deg": 35.0, "raz_deg": 10.0},
            "rtol": {"toa": 1e-3, "boa": 2e-3},
        },
        "atmosphere": _solar_atmos(),
        # mapping per case_C_solar_tester_vector.jl:24-50
        "greek": {"alpha": a2, "beta": a1, "gamma": b1, "delta": a4, "epsilon": [-x for x in b2], "zeta": a3},
        "truth": {s: _reshape4(_vector(t, "SOLAR_TESTER_VECTOR_" + s)) for s in ("I", "Q", "U")},
    }
    return "solar_tester_vector.json", fx


def _jld2_arrays(path):
    """Minimal reader for the HDF5 subset JLD2 writes: version-2 object headers (OHDR), link messages in the root
    group, contiguous little-endian float datasets.  Addresses are relative to the 512-byte superblock offset."""
    import struct
    d = open(path, "rb").read()
    base = 512

    def msgs(off):
        assert d[off:off + 4] == b"OHDR"
        flags = d[off + 5]
        p = off + 6 + (16 if flags & 0x20 else 0) + (4 if flags & 0x10 else 0)
        szb = 1 << (flags & 3)
        size = int.from_bytes(d[p:p + szb], "little")
        p += szb
        end, out = p + size, []
        while p < end - 4:
            t, sz = d[p], struct.unpack("<H", d[p + 1:p + 3])[0]
            p += 4 + (2 if flags & 0x04 else 0)
            out.append((t, d[p:p + sz]))
            p += sz
        return out

    links = {}
    for m in re.finditer(b"OHDR", d):
        for t, b in msgs(m.start()):
            if t == 6:  # link message
                fl, q = b[1], 2
                lt = 0
                if fl & 0x08:
                    lt, q = b[q], q + 1
                q += (8 if fl & 0x04 else 0) + (1 if fl & 0x10 else 0)
                ls = 1 << (fl & 3)
                ln = int.from_bytes(b[q:q + ls], "little")
                q += ls
                name = b[q:q + ln].decode()
                if lt == 0:
                    links[name] = struct.unpack("<Q", b[q + ln:q + ln + 8])[0]
    out = {}
    for name, addr in links.items():
        dims = dt = lay = None
        for t, b in msgs(base + addr):
            if t == 1:
                dims = struct.unpack("<%dQ" % b[1], b[4:4 + 8 * b[1]])
            elif t == 3:
                dt = (b[0] & 15, struct.unpack("<I", b[4:8])[0])
            elif t == 8 and b[1] == 1:
                lay = struct.unpack("<QQ", b[2:18])
        if dims and len(dims) > 0 and dt and dt[0] == 1 and lay:
            a = np.frombuffer(d[base + lay[0]:base + lay[0] + lay[1]], dtype="<f%d" % dt[1])
            out[name] = a.reshape(dims)   # HDF5 (row-major) dims = reversed Julia dims
    return out


def raman_phase1b():
    import yaml
    arrs = _jld2_arrays(os.path.join(REF, "test/reference/phase1b_RRS_sanghavi_q0.jld2"))
    with open(os.path.join(REF, "test/test_parameters/Phase1b_RRS_761-764nm.yaml"), encoding="utf-8") as f:
        y = yaml.safe_load(f)
    rt, geo, atm = y["radiative_transfer"], y["geometry"], y["atmospheric_profile"]
    assert rt["spec_bands"] == ["(1e7/765):0.5:(1e7/762)"], rt["spec_bands"]
    fx = dict(
        source="test/reference/phase1b_RRS_sanghavi_q0.jld2 + test/test_parameters/Phase1b_RRS_761-764nm.yaml",
        procedure="test/test_forward_raman_phase1b.jl:41-103: Stokes_IQU, nstreams 3, Float32, depol auto, Lambertian "
                  "albedo 0, F0 = e1; n2/o2 at the column-mean temperature; getRamanSSProp!(RS, 1e7/mean(nu), nu); "
                  "compare R, T, ieR, ieT [nVZA, nStokes, nSpec] with atol 1e-6, rtol 0.02 (I, Q) and atol 1e-6 (U)",
        nu_start=1e7 / 765, nu_step=0.5, nu_stop=1e7 / 762,
        nstreams=int(rt["nstreams"]), depol=float(rt["depol"]), polarization="IQU", albedo=0.0,
        sza=float(geo["sza"]), vza=[float(v) for v in geo["vza"]], vaz=[float(v) for v in geo["vaz"]],
        T=[float(v) for v in atm["T"]], p=[float(v) for v in atm["p"]], q=[float(v) for v in atm["q"]],
        profile_reduction=int(atm["profile_reduction"]),
        atol=1e-6, rtol=0.02,
    )
    for k in ("R_rrs", "T_rrs", "ieR", "ieT"):
        a = arrs[k]                      # [nSpec, nStokes, nVZA] in file order
        fx[k] = np.transpose(a, (2, 1, 0)).astype(float).tolist()
    return "phase1b_rrs_sanghavi_q0.json", fx


def water_table():
    """The three literal columns of src/CoreRT/Surfaces/water_refraction.jl:15-57 (published Segelstein 1981 values)."""
    with open(os.path.join(REF, "src/CoreRT/Surfaces/water_refraction.jl"), encoding="utf-8") as f:
        text = _strip_comments(f.read())
    fx = dict(source="src/CoreRT/Surfaces/water_refraction.jl:15-57 (Segelstein 1981)",
              wavelength_nm=_vector(text, "_WATER_RI_TABLE_NM"), n_real=_vector(text, "_WATER_N_REAL"),
              k_imag=_vector(text, "_WATER_K_IMAG"))
    assert len(fx["wavelength_nm"]) == len(fx["n_real"]) == len(fx["k_imag"]) == 92
    return "segelstein1981_water.json", fx


def ocean_coxmunk_scene():
    """BASELINE config C3: the scene parameters of config/ocean_coxmunk.yaml (parsed values: band, surface constructor,
    streams, geometry, 33-layer profile) -- the input the linearized Cox-Munk parity test runs on."""
    import yaml
    with open(os.path.join(REF, "config/ocean_coxmunk.yaml"), encoding="utf-8") as f:
        d = yaml.safe_load(f)
    d["source"] = "config/ocean_coxmunk.yaml (parsed with yaml.safe_load)"
    assert len(d["atmospheric_profile"]["T"]) == 33 and len(d["atmospheric_profile"]["p"]) == 34
    return "ocean_coxmunk_scene.json", d


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not found at %s (fixtures are committed; nothing to do)" % REF)
    for fn in (siewert, natraj, sixsv, solar_scalar, solar_vector, raman_phase1b, water_table, ocean_coxmunk_scene):
        name, fx = fn()
        with open(os.path.join(OUT, name), "w") as f:
            json.dump(fx, f, indent=1)
        print("wrote", name, os.path.getsize(os.path.join(OUT, name)), "bytes")


if __name__ == "__main__":
    main()
