#!/usr/bin/env python
"""Write a synthetic stand-in for the IceCube 3-year public MC file
(`events/IceCube_3y_oscillations/neutrino_mc.csv.bz2`, which is not shipped with
the reference either -- it has to be downloaded, README of that directory) in the
same CSV layout that data.csv_loader reads (pisa/stages/data/csv_loader.py:108-166
and IceCube_3y_neutrinos.cfg: columns pdg, type, true_energy, true_coszen, weight,
reco_energy, reco_coszen, pid).

    python scripts/make_synthetic_3y_mc.py OUT_DIR [N_EVENTS] [SEED]

writes OUT_DIR/events/IceCube_3y_oscillations/neutrino_mc.csv.bz2; put OUT_DIR on
$PISA_RESOURCES to run settings/pipeline/IceCube_3y_neutrinos.cfg unmodified.
"""
import os
import sys

import numpy as np
import pandas as pd


def make(n_events=120000, seed=0):
    rs = np.random.RandomState(seed)
    pdg = rs.choice([12, 14, 16, -12, -14, -16], size=n_events, p=[0.2, 0.3, 0.1, 0.15, 0.2, 0.05])
    itype = rs.choice([0, 1, 2], size=n_events, p=[0.25, 0.6, 0.15])  # 0 = NC, >= 1 = CC
    true_energy = 10 ** (rs.rand(n_events) * 2.6 + 0.2)               # 1.6 .. 630 GeV
    true_coszen = rs.rand(n_events) * 2 - 1
    reco_energy = true_energy * np.exp(rs.normal(0, 0.25, n_events))
    reco_coszen = np.clip(true_coszen + rs.normal(0, 0.2, n_events), -1, 1)
    pid = (rs.rand(n_events) < np.where(np.abs(pdg) == 14, 0.6, 0.25)).astype(float)
    weight = 1e-6 * true_energy ** 0.6 * (0.5 + rs.rand(n_events))   # "weighted_aeff" column
    return pd.DataFrame(dict(pdg=pdg, type=itype, true_energy=true_energy, true_coszen=true_coszen,
                             weight=weight, reco_energy=reco_energy, reco_coszen=reco_coszen, pid=pid))


if __name__ == "__main__":
    out_dir = sys.argv[1]
    n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 120000
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    path = os.path.join(out_dir, "events", "IceCube_3y_oscillations")
    os.makedirs(path, exist_ok=True)
    make(n, seed).to_csv(os.path.join(path, "neutrino_mc.csv.bz2"), index=False)
    print(os.path.join(path, "neutrino_mc.csv.bz2"))
