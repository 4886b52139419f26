"""ProductManager: YAML configuration -> telescope / BeamTransfer / KL objects and the
order in which products are generated (drift/core/manager.py:28-305)."""
import logging
import os

import yaml

from . import beamtransfer, cylinder, doublekl, kltransform, parallel, psestimation, storage

logger = logging.getLogger(__name__)

teltype_dict = {
    "UnpolarisedCylinder": cylinder.UnpolarisedCylinderTelescope,
    "PolarisedCylinder": cylinder.PolarisedCylinderTelescope,
}

kltype_dict = {"KLTransform": kltransform.KLTransform, "DoubleKL": doublekl.DoubleKL}
# The reference's Monte-Carlo estimators ("MonteCarlo", "MonteCarloAlt") estimate the same Fisher matrix from
# random realisations; here they resolve to the exact computation (the expectation they converge to), with a warning.
pstype_dict = {"Full": psestimation.PSExact, "MonteCarlo": psestimation.PSExact, "MonteCarloAlt": psestimation.PSExact}



def _resolve_class(clstype, clsdict, objtype=""):
    if isinstance(clstype, dict):
        import importlib
        import importlib.util

        if "file" in clstype:
            spec = importlib.util.spec_from_file_location(clstype["module"], clstype["file"])
            module = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(module)
        else:
            module = importlib.import_module(clstype["module"])
        return getattr(module, clstype["class"])
    if clstype in clsdict:
        return clsdict[clstype]
    raise Exception("Unsupported %s" % objtype)


class ProductManager(object):
    directory = None
    gen_beams = False
    gen_kl = False
    gen_ps = False
    skip_svd = False
    skip_svd_inv = False

    @classmethod
    def from_config(cls, configfile):
        configfile = os.path.normpath(os.path.expandvars(os.path.expanduser(configfile)))
        if not os.path.exists(configfile):
            raise Exception("Configuration file does not exist %s." % configfile)
        if os.path.isdir(configfile):
            configfile = configfile + "/config.yaml"
        with open(configfile, "r") as f:
            yconf = yaml.safe_load(f)
        outdir = yconf["config"]["output_directory"]
        dfile = os.path.join(outdir, "config.yaml")
        if parallel.io_root():
            if not os.path.isabs(outdir):
                outdir_abs = os.path.abspath(os.path.normpath(os.path.join(os.path.dirname(configfile), outdir)))
            else:
                outdir_abs = outdir
            os.makedirs(outdir_abs, exist_ok=True)
            dfile = os.path.join(outdir_abs, "config.yaml")
            if not os.path.exists(dfile) or not os.path.samefile(configfile, dfile):
                with open(configfile, "r") as f:
                    contents = f.read()
                if outdir_abs != outdir:
                    contents = contents.replace(outdir, outdir_abs)
                with open(dfile, "w+") as f:
                    f.write(contents)
        dfile = parallel.bcast_object(dfile)
        parallel.barrier()
        c = cls()
        with open(dfile) as f:
            c.apply_config(yaml.safe_load(f))
        return c

    def apply_config(self, yconf):
        if "config" not in yconf:
            raise ValueError("Configuration file must have an 'config' section.")
        if "telescope" not in yconf:
            raise ValueError("Configuration file must have an 'telescope' section.")
        self.config = yconf
        self.directory = os.path.expandvars(os.path.expanduser(yconf["config"]["output_directory"]))
        telclass = _resolve_class(yconf["telescope"]["type"], teltype_dict, "telescope")
        self.telescope = telclass.from_config(yconf["telescope"])
        btclass = beamtransfer.BeamTransfer
        if yconf["config"].get("nosvd"):
            btclass = beamtransfer.BeamTransferNoSVD
        if yconf["config"].get("fullsvd"):
            btclass = beamtransfer.BeamTransferFullSVD
        self.beamtransfer = btclass(self.directory + "/bt/", telescope=self.telescope)
        self.beamtransfer.read_config(yconf["config"])
        self.gen_beams = bool(yconf["config"].get("beamtransfers"))
        self.skip_svd = bool(yconf["config"].get("skip_svd"))
        self.kltransforms = {}
        for klentry in yconf.get("kltransform", []) or []:
            klclass = _resolve_class(klentry["type"], kltype_dict, "KL filter")
            self.kltransforms[klentry["name"]] = klclass.from_config(klentry, self.beamtransfer, subdir=klentry["name"])
        self.gen_kl = bool(yconf["config"].get("kltransform"))
        # power-spectrum estimators (manager.py:251-277)
        self.psestimators = {}
        self.gen_ps = bool(yconf["config"].get("psfisher"))
        if self.gen_ps and "psfisher" not in yconf:
            raise Exception("Require a psfisher section if config: psfisher is Yes.")
        for psentry in yconf.get("psfisher", []) or []:
            psclass = _resolve_class(psentry["type"], pstype_dict, "PS estimator")
            if isinstance(psentry["type"], str) and psentry["type"].startswith("MonteCarlo"):
                logger.warning("psfisher type %s: computing the exact Fisher matrix instead of a Monte-Carlo estimate"
                               % psentry["type"])
            klname = psentry["klname"]
            psname = psentry.get("name", "ps")
            if klname not in self.kltransforms:
                import warnings

                warnings.warn("Desired KL object (name: %s) does not exist." % klname)
                self.psestimators[psname] = None
            else:
                self.psestimators[psname] = psclass.from_config(psentry, self.kltransforms[klname], subdir=psname)

        if self.gen_kl and self.kltransforms and "kl_cost_weight" not in yconf["config"]:
            # the cost of an m-block downstream of the SVD chain, for the m-ranges of the ranks: a DoubleKL is two
            # eigenproblems plus, at low m, the non-positive-definite rescue; a Fisher estimator projects every band
            # (round 5: 0.19 / 0.57 / 0.14 in the units of `BeamTransfer._m_cost` — the SVD chain got a third cheaper
            # this round and the stages behind it did not: the eight configs[3] shares measured with the round-4 weights
            # (0.19 / 0.35 / 0.085) ran 28.5 .. 25.2 s from rank 0 to rank 7, and their densities re-partitioned ask for a
            # total of 0.90, profiles/r05z_configs3_shares.json; round 3 had 0.25 / 0.625 / 0.125)
            w = sum(0.57 if isinstance(k, doublekl.DoubleKL) else 0.19 for k in self.kltransforms.values())
            if self.gen_ps:
                w += 0.14 * sum(1 for p in self.psestimators.values() if p is not None)
            self.beamtransfer.kl_cost_weight = float(w)

    def generate(self):
        os.makedirs(self.directory, exist_ok=True)
        if parallel.io_root():
            with open(os.path.join(self.directory, "configdump.yaml"), "w") as fh:
                dump = dict(self.config)
                # the settings of the SHT the products are made with, resolved (defaults included): healpy's `iter` and
                # ring weights reach the reference through cora and cannot be read here
                tel = self.telescope
                dump["driftscan_amd"] = dict(sht_iter=int(getattr(tel, "sht_iter", 0) or 0),
                                             sht_ring_weights=bool(getattr(tel, "sht_ring_weights", None) is not None))
                yaml.dump(dump, fh)
        # manager.py:278-305 runs the stages one after the other through the files.  Here the KL transforms of a batch of
        # m run right behind its SVD chain, while the SVD products are resident in HBM (BeamTransfer.generate's
        # `after_batch`); `klobj.generate()` then only finishes what is left (nothing, unless the beams existed already),
        # waits for the writers and collects the spectra.
        kls = list(self.kltransforms.values()) if self.gen_kl else []
        pss = [p for p in self.psestimators.values() if p is not None] if (self.gen_ps and self.gen_kl) else []

        if self.gen_beams:
            for psobj in pss:   # the estimators take the modes of a batch from memory, right behind its KL transform
                psobj.kltrans.__dict__.setdefault("_mode_cache", {})

            def after_batch(ms):
                for klobj in kls:
                    klobj.generate_ms(ms)
                for psobj in pss:
                    psobj.accumulate_ms(ms)
                for klobj in kls:
                    if "_mode_cache" in klobj.__dict__:
                        klobj.__dict__["_mode_cache"].clear()

            resident = bool(kls) and not self.skip_svd
            if resident:
                # every rank opens its local Fisher sums NOW — also one whose range of m is empty and that never sees a
                # batch: `PSEstimation.generate` decides between "my range of the resident pipeline" and the file-based
                # partition by the presence of these sums, and that choice has to be the same on all ranks
                for psobj in pss:
                    psobj.accumulate_ms([])
            self.beamtransfer.generate(skip_svd=self.skip_svd, after_batch=after_batch if resident else None)
        if self.gen_kl:
            for klname, klobj in self.kltransforms.items():
                klobj.generate()
        for klobj in kls:
            klobj.__dict__.pop("_mode_cache", None)
        storage.trace_mark("beam transfer and KL products generated")
        if self.gen_ps:
            for psname, psobj in self.psestimators.items():
                if psobj is None:
                    continue
                psobj.generate()
                psobj.delbands()
        if parallel.rank0():
            logger.info("DONE GENERATING PRODUCTS")
