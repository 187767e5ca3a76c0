"""Generates tests/golden/* from the reference's own fixtures.  Runs ONLY in the build container
(needs /root/reference); the outputs are committed, this script documents how they were made.

  * copies (as data) the reference's thermalised gauge configurations used by its dynamical tests
    (test/test_wilson.toml:17, test/test_staggered.toml:15, test/test_domainwallhmc.toml:17)
  * decodes every SU(3) fixture with the product's readers and records plaquette / unitarity computed
    by the oracle -> golden.json  (cross-checked against SURVEY.md Appendix B)
"""
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import latticeqcd_jl_amd as lq  # noqa: E402
from oracle import oracle as orc  # noqa: E402

REF = "/root/reference/test"
SU3 = {
    "confs_HMC_L04040404_beta5.7_Wilson_kappa0.141139": (4, 4, 4, 4),
    "confs_HMC_L04040404_beta5.7_Staggered_mass0.5": (4, 4, 4, 4),
    "confs_HMC_L04040404_beta5.7_Staggered_mass0.5_Nf2": (4, 4, 4, 4),
    "confs_HMC_L04040404_beta5.7_Staggered_mass0.5_Nf3": (4, 4, 4, 4),
    "confs_HMC_L04040404_beta5.7_Domainwall": (4, 4, 2, 2),
    "confs_HMC_L04040404_beta5.7_quenched_su3": (4, 4, 4, 4),
    "confs_Heatbath_L04040404_beta5.7_quenched_su3": (4, 4, 4, 4),
}
COPY = {
    "confs_HMC_L04040404_beta5.7_Wilson_kappa0.141139/conf_00000100.ildg": "wilson_4x4x4x4.ildg",
    "confs_HMC_L04040404_beta5.7_Staggered_mass0.5/conf_00000100.ildg": "staggered_4x4x4x4.ildg",
    "confs_HMC_L04040404_beta5.7_Staggered_mass0.5_Nf2/conf_00000100.ildg": "staggered_nf2_4x4x4x4.ildg",
    "confs_HMC_L04040404_beta5.7_Staggered_mass0.5_Nf3/conf_00000100.ildg": "staggered_nf3_4x4x4x4.ildg",
    "confs_HMC_L04040404_beta5.7_quenched_su3/conf_00000100.ildg": "quenched_su3_4x4x4x4.ildg",
    "confs_HMC_L04040404_beta5.7_Domainwall/conf_00000100.ildg": "domainwall_4x4x2x2.ildg",
    "confs_HMC_L04040404_beta5.7_Domainwall/conf_00000100.ildg.txt": "domainwall_4x4x2x2.ildg.txt",
}

gold = {"plaquette": {}, "unitarity_dev": {}, "ildg_vs_text_maxabs": {}, "L": {}}
for d, L in SU3.items():
    base = os.path.join(REF, d, "conf_00000100.ildg")
    Ub = lq.gauge_io.load_ildg(base, L)
    Ut = lq.gauge_io.load_BridgeText(base + ".txt", L)
    gold["L"][d] = list(L)
    gold["plaquette"][d] = orc.plaquette(Ub, L)
    gold["unitarity_dev"][d] = orc.unitarity_dev(Ub, L)
    gold["ildg_vs_text_maxabs"][d] = float(np.abs(Ub - Ut).max())
    print(d, L, gold["plaquette"][d], gold["unitarity_dev"][d], gold["ildg_vs_text_maxabs"][d])

# the reference's loose end-to-end goldens (test/debugplaqdata.txt:7-11), recorded for a future Julia-hosted run
gold["reference_end_of_run_plaquettes_10pct"] = {
    "quenched_su3_hmc": 0.55783720583739, "wilson": 0.5784043949012552, "staggered_nf4": 0.5734383856968012, "staggered_nf2": 0.56287171870089,
    "staggered_nf3": 0.5595757232711884, "domainwall": 0.5757839405690621}
for src, dst in COPY.items():
    shutil.copyfile(os.path.join(REF, src), os.path.join(HERE, dst))
    os.chmod(os.path.join(HERE, dst), 0o644)
with open(os.path.join(HERE, "golden.json"), "w") as f:
    json.dump(gold, f, indent=1, sort_keys=True)
