"""Events from an SQLite database (counterpart of pisa/stages/data/sqlite_loader.py:17-149): per output name the rows
of table `truth` with its (signed) PDG code and interaction type, joined by `event_no` with `reconstruction` (or
`retro` for post_fix '_retro'); columns true_energy / true_coszen (= cos zenith), reco_energy / reco_coszen / pid from
the `<name><post_fix>` columns, unit weights and `weighted_aeff = 1e-4 OneWeight / n_files / gen_ratio / NEvents`
with n_files the number of distinct (RunID, SubrunID) of that PDG code."""
import sqlite3

import numpy as np

from pisa_amd import FTYPE
from pisa_amd.core.container import Container
from pisa_amd.core.stage import Stage

__all__ = ["sqlite_loader"]


class sqlite_loader(Stage):  # pylint: disable=invalid-name
    def __init__(self, database, output_names, post_fix="_pred", **std_kwargs):
        self.database = database
        self.post_fix = post_fix
        super().__init__(expected_params=(), expected_container_keys=(), **std_kwargs)
        self.output_names = output_names

    def get_pid_and_interaction_type(self, name):
        """(signed PDG code, interaction type, nubar, flavour) from a container name; as in the reference the LAST
        of the tags 'e', 'mu', 'tau' found in the name decides (sqlite_loader.py:49-68)"""
        nubar = -1 if "bar" in name else 1
        for tag, code, flav in (("e", 12, 0), ("mu", 14, 1), ("tau", 16, 2)):
            if tag in name:
                pid, flavor = code, flav
        for tag, code in (("cc", 1), ("nc", 2)):
            if tag in name:
                interaction_type = code
        return nubar * pid, interaction_type, nubar, flavor

    def query_database(self, interaction_type, pid):
        import pandas as pd

        with sqlite3.connect(self.database) as con:
            truth = pd.read_sql("SELECT * FROM truth WHERE interaction_type = %s and pid = %s" % (interaction_type, pid), con)
            truth = truth.sort_values("event_no").reset_index(drop=True)
            table = "retro" if self.post_fix == "_retro" else "reconstruction"
            reco = pd.read_sql("SELECT * FROM %s WHERE event_no in %s" % (table, str(tuple(truth["event_no"]))), con)
            reco = reco.sort_values("event_no").reset_index(drop=True)
            n_files = len(pd.read_sql("SELECT DISTINCT RunID, SubrunID FROM truth WHERE pid = %s" % pid, con))
        return truth, reco, n_files

    def setup_function(self):
        for name in self.output_names:
            container = Container(name)
            pid, interaction_type, nubar, flavor = self.get_pid_and_interaction_type(name)
            truth, reco, n_files = self.query_database(interaction_type, pid)
            container["true_coszen"] = np.cos(truth["zenith"]).values.astype(FTYPE)
            container["true_energy"] = truth["energy"].values.astype(FTYPE)
            container.set_aux_data("nubar", nubar)
            container.set_aux_data("flav", flavor)
            container["reco_coszen"] = np.cos(reco["zenith" + self.post_fix]).values.astype(FTYPE)
            container["reco_energy"] = reco["energy" + self.post_fix].values.astype(FTYPE)
            pid_column = "L7_PIDClassifier_FullSky_ProbTrack" if self.post_fix == "_retro" else "track" + self.post_fix
            container["pid"] = reco[pid_column].values.astype(FTYPE)
            container["weights"] = np.ones(container.size, dtype=FTYPE)
            container["initial_weights"] = np.ones(container.size, dtype=FTYPE)
            weighted_aeff = 1e-4 * truth["OneWeight"] / n_files / truth["gen_ratio"] / truth["NEvents"]
            container["weighted_aeff"] = weighted_aeff.values.astype(FTYPE)
            self.data.add_container(container)
        if len(self.data.names) == 0:
            raise ValueError("No containers created during data loading for some reason.")

    def apply_function(self):
        for container in self.data:
            container["weights"] = np.copy(container["initial_weights"])


def write_test_database(path, n_evts=10, seed=42):
    """the ten-event database of the reference's `init_test` (sqlite_loader.py:152-177)"""
    rs = np.random.RandomState(seed)
    true_data, reco_data = [], []
    for i in range(n_evts):
        true_data.append(tuple(list(rs.random_sample(4).astype(float)) + [i, n_evts, 1, 14, 1, 0]))
        reco_data.append(tuple(list(rs.random_sample(3).astype(float)) + [i]))
    with sqlite3.connect(path) as con:
        cur = con.cursor()
        cur.execute("CREATE TABLE truth(energy, zenith, OneWeight, gen_ratio, event_no, NEvents, interaction_type, pid, RunID,"
                    " SubrunID)")
        cur.executemany("INSERT INTO truth VALUES(?, ?, ?, ?, ?, ?, ?, ?, ?, ?)", true_data)
        cur.execute("CREATE TABLE reconstruction(energy_pred, zenith_pred, track_pred, event_no)")
        cur.executemany("INSERT INTO reconstruction VALUES(?, ?, ?, ?)", reco_data)
    return true_data, reco_data


def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    import os
    import tempfile

    path = os.path.join(tempfile.gettempdir(), "pisa_amd_sqlite_loader_test_file_%d" % os.getpid())
    if not os.path.isfile(path):
        write_test_database(path)
    return sqlite_loader(database=path, output_names=["numu_cc"])
