"""Host-side mirror of the reference's call sequence for the pairwise path, on device-resident data.

Names follow the reference's seams so that tests read like its call sites:
  get_vectors            <- CoETools::getVectors (CoMap/CoETools.cpp:366-416) + writeInfos columns (:507-510)
  compute_norms          <- AnalysisTools::computeNorms (CoMap/AnalysisTools.cpp:343-350) (fused into the mapping kernel)
  null_distribution      <- AnalysisTools::getNullDistributionIntraDR (CoMap/AnalysisTools.cpp:564-658)
  compute_intra_stats    <- CoETools::computeIntraStats (CoMap/CoETools.cpp:604-728), dense outputs instead of TSV rows
  compute_intra_rows     <- the same pair loop for a range of rows i, compacted to the rows of statistics.txt on the device
                            with no N x N matrix (what a rank of a multi-GPU job runs on its share of the upper triangle)
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
        self._dense = None
        self._rows = None
        self._compact = None
        self._null = {}
        self._null_aln = None

    def _dense_buffers(self):
        if self._dense is None:
            dev = self.aln.device
            self._dense = (torch.empty((self.n, self.n), dtype=torch.float64, device=dev),
                           torch.empty((self.n, self.n), dtype=torch.float64, device=dev),
                           torch.empty((self.n, self.n), dtype=torch.int32, device=dev))
        return self._dense

    @property
    def stat(self):
        return self._dense_buffers()[0]

    @property
    def pvalue(self):
        return self._dense_buffers()[1]

    @property
    def nsim(self):
        return self._dense_buffers()[2]

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

    def null_distribution(self, seed, rep_begin, rep_end, rep_ram, supplied=None, map_events=None, sim_events=None):
        """AnalysisTools::getNullDistributionIntraDR for replicates [rep_begin, rep_end).  map_events = (start, end) CUDA
        events: the alignments are simulated into a buffer of this object first (cmx_null_simulate_dev) and the events
        bracket the mapping launch alone -- what bench.py's roofline is about (sim_events bracket the simulator, so that
        the two can be read side by side); otherwise one call does both."""
        b = self.null_buffers((rep_end - rep_begin) * rep_ram)
        if map_events is not None and supplied is None:
            nbytes = (rep_end - rep_begin) * 2 * self.eng.T * rep_ram
            if self._null_aln is None or self._null_aln.numel() < nbytes:
                self._null_aln = torch.empty(nbytes, dtype=torch.uint8, device=self.aln.device)
            if sim_events is not None:
                sim_events[0].record()
            self.eng.null_simulate_dev(seed, rep_begin, rep_end, rep_ram, self._null_aln)
            if sim_events is not None:
                sim_events[1].record()
            supplied = self._null_aln
        if map_events is not None:
            map_events[0].record()
        self.eng.null_intra_dev(self.kind, seed, rep_begin, rep_end, rep_ram, b["stat"], b["rcmin"], b["prmin"],
                                b["nmin"], supplied=supplied, threshold=self.threshold)
        if map_events is not None:
            map_events[1].record()
        return b

    def compute_intra_stats(self, null_stat=None, null_nmin=None):
        self.eng.pair_stats_dev(self.kind, self.counts, self.stat, threshold=self.threshold)
        if null_stat is not None:
            self.eng.intra_pvalues_dev(self.stat, self.norm, self.nclasses, null_stat, null_nmin, self.pvalue, self.nsim)
        return self.stat, self.pvalue, self.nsim

    def compute_intra_rows(self, null_stat=None, null_nmin=None, row_begin=0, row_end=None, filters=None, capacity=None):
        """-> (rows uint8 CUDA tensor viewable as engine.PAIR_ROW records, count int64 CUDA tensor [1])"""
        from .engine import PAIR_ROW
        row_end = self.n if row_end is None else row_end
        # pairs (i, j > i) with row_begin <= i < row_end
        npairs = sum_pairs(self.n, row_begin, row_end)
        cap = npairs if capacity is None else int(capacity)
        if self._rows is None or self._rows[0].numel() < max(cap, 1) * PAIR_ROW.itemsize:
            dev = self.aln.device
            self._rows = (torch.empty(max(cap, 1) * PAIR_ROW.itemsize, dtype=torch.uint8, device=dev),
                          torch.zeros(1, dtype=torch.int64, device=dev))
        rows, count = self._rows
        self.eng.intra_rows_range_dev(self.kind, self.counts, self.rate_class, self.post_rate, self.norm, null_stat, null_nmin,
                                      self.nclasses, rows, count, row_begin, row_end, filters, threshold=self.threshold)
        return rows, count


    def prefetch_intra_gram(self, row_begin=0, row_end=None):
        """The statistics of the observed pairs do not depend on the null: a caller that maps the observed alignment on a side
        stream can enqueue them there too, behind get_vectors; compute_intra_compact with the same row range then only looks
        the p-values up and writes the records (its stream ordered behind this one's)."""
        self.eng.intra_gram_prefetch_dev(self.kind, self.counts, self.n, row_begin, self.n if row_end is None else row_end)

    def compute_intra_compact(self, null_stat=None, null_nmin=None, row_begin=0, row_end=None):
        """the unfiltered pair loop as 16-byte records (engine.PAIR_COMPACT) -> (uint8 CUDA tensor, number of pairs)"""
        from .engine import PAIR_COMPACT
        row_end = self.n if row_end is None else row_end
        npairs = sum_pairs(self.n, row_begin, row_end)
        if self._compact is None or self._compact.numel() < max(npairs, 1) * PAIR_COMPACT.itemsize:
            self._compact = torch.empty(max(npairs, 1) * PAIR_COMPACT.itemsize, dtype=torch.uint8, device=self.aln.device)
        self.eng.intra_compact_range_dev(self.kind, self.counts, self.norm, null_stat, null_nmin, self.nclasses, self._compact,
                                         row_begin, row_end, threshold=self.threshold)
        return self._compact, npairs


def sum_pairs(n, row_begin, row_end):
    """number of pairs (i, j) with row_begin <= i < row_end and i < j < n"""
    a, b = int(row_begin), int(row_end)
    return (b - a) * (n - 1) - (b * (b - 1) - a * (a - 1)) // 2
