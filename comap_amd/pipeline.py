"""Host-side mirror of the reference's call sequence for the pairwise path, on device-resident data.

Names follow the reference's seams so that tests read like its call sites:
  get_vectors            <- CoETools::getVectors (CoMap/CoETools.cpp:366-416) + writeInfos columns (:507-510)
  compute_norms          <- AnalysisTools::computeNorms (CoMap/AnalysisTools.cpp:343-350) (fused into the mapping kernel)
  null_distribution      <- AnalysisTools::getNullDistributionIntraDR (CoMap/AnalysisTools.cpp:564-658)
  compute_intra_stats    <- CoETools::computeIntraStats (CoMap/CoETools.cpp:604-728), dense outputs instead of TSV rows
torch is used for device memory, streams and (in bench.py) torch.distributed only; every number is produced by the
HIP kernels behind the C-ABI (comap_amd/engine.py)."""
import torch


class IntraAnalysis:
    def __init__(self, engine, d_aln, statistic="Correlation", nclasses=10, threshold=0.99):
        from .engine import STAT_BY_NAME
        assert d_aln.is_cuda and d_aln.dtype == torch.uint8 and d_aln.dim() == 2
        self.eng = engine
        self.aln = d_aln
        self.kind = STAT_BY_NAME[statistic] if isinstance(statistic, str) else int(statistic)
        self.nclasses = int(nclasses)        # statistic.null.nb_rate_classes (CoETools.cpp:638)
        self.threshold = float(threshold)
        self.n = d_aln.shape[1]
        dev = d_aln.device
        f64 = dict(dtype=torch.float64, device=dev)
        self.counts = torch.empty((engine.B * engine.K, self.n), **f64)
        self.logL = torch.empty(self.n, **f64)
        self.post_rate = torch.empty(self.n, **f64)
        self.norm = torch.empty(self.n, **f64)
        self.rate_class = torch.empty(self.n, dtype=torch.int32, device=dev)
        self.stat = torch.empty((self.n, self.n), **f64)
        self.pvalue = torch.empty((self.n, self.n), **f64)
        self.nsim = torch.empty((self.n, self.n), dtype=torch.int32, device=dev)
        self._null = {}

    def get_vectors(self, aln=None):
        self.eng.map_sites_dev(self.aln if aln is None else aln, self.counts, self.logL, self.post_rate, self.rate_class, self.norm)
        return self.counts

    def compute_norms(self):
        return self.norm

    def null_buffers(self, n):
        if self._null.get("n") != n:
            dev = self.aln.device
            self._null = dict(n=n, stat=torch.empty(n, dtype=torch.float64, device=dev),
                              nmin=torch.empty(n, dtype=torch.float64, device=dev),
                              prmin=torch.empty(n, dtype=torch.float64, device=dev),
                              rcmin=torch.empty(n, dtype=torch.int32, device=dev))
        return self._null

    def null_distribution(self, seed, rep_begin, rep_end, rep_ram, supplied=None):
        b = self.null_buffers((rep_end - rep_begin) * rep_ram)
        self.eng.null_intra_dev(self.kind, seed, rep_begin, rep_end, rep_ram, b["stat"], b["rcmin"], b["prmin"],
                                b["nmin"], supplied=supplied, threshold=self.threshold)
        return b

    def compute_intra_stats(self, null_stat=None, null_nmin=None):
        self.eng.pair_stats_dev(self.kind, self.counts, self.stat, threshold=self.threshold)
        if null_stat is not None:
            self.eng.intra_pvalues_dev(self.stat, self.norm, self.nclasses, null_stat, null_nmin, self.pvalue, self.nsim)
        return self.stat, self.pvalue, self.nsim
