"""Empirical substitution-model tables used to build the generator Q handed to the engine.

The engine itself (C-ABI, ``include/comap_mi355x.h``) takes Q, pi, rates and weights as
plain arrays; nothing in this file is on the device path.  The reference obtains these
tables from Bio++ (``model = JTT92`` in examples/Proteins/Benchmark/CoMap/comap.bpp:40,
which Bio++ implements with the DCmut estimate of Kosiol & Goldman 2005); Bio++ is not in
the reference tree, so the table below is typed in from the literature (PAML's
``jones-dcmut.dat`` layout: lower-triangular exchangeabilities, then frequencies; amino
acid order A R N D C Q E G H I L K M F P S T W Y V).  It is validated by reproducing the
reference's committed ``Myo.infos`` / ``Myo_unif.vec`` fixtures (tests/test_golden_myoglobin.py).
"""
import numpy as np

AA_ORDER = "ARNDCQEGHILKMFPSTWYV"

_JTT_DCMUT_LOWER = """
0.531678
0.557967 0.451095
0.827445 0.154899 5.549530
0.574478 1.019843 0.313311 0.105625
0.556725 3.021995 0.768834 0.521646 0.091304
1.066681 0.318483 0.578115 7.766557 0.053907 3.417706
1.740159 1.359652 0.773313 1.272434 0.546389 0.231294 1.115632
0.219970 3.210671 4.025778 1.032342 0.724998 5.684080 0.243768 0.201696
0.361684 0.239195 0.491003 0.115968 0.150559 0.078270 0.111773 0.053769 0.181788
0.310007 0.372261 0.137289 0.061486 0.164593 0.709004 0.097485 0.069492 0.540571 2.335139
0.369437 6.529255 2.529517 0.282466 0.049009 2.966732 1.731684 0.269840 0.525096 0.202562 0.146481
0.469395 0.431045 0.330720 0.190001 0.409202 0.456901 0.175084 0.130379 0.329660 4.831666 3.856906 0.624581
0.138293 0.065314 0.073481 0.032522 0.678335 0.045683 0.043829 0.050212 0.453428 0.777090 2.500294 0.024521 0.436181
1.959599 0.710489 0.121804 0.127164 0.123653 1.608126 0.191994 0.208081 1.141961 0.098580 1.060504 0.216345 0.164215 0.148483
3.887095 1.001551 5.057964 0.589268 2.155331 0.548807 0.312449 1.874296 0.743458 0.405119 0.592511 0.474478 0.285564 0.943971 2.788406
4.582565 0.650282 2.351311 0.425159 0.469823 0.523825 0.331584 0.316862 0.477355 2.553806 0.272514 0.965641 2.114728 0.138904 1.176961 4.777647
0.084329 1.257961 0.027700 0.057466 1.104181 0.172206 0.114381 0.544180 0.128193 0.134510 0.530324 0.089134 0.201334 0.537922 0.069965 0.310927 0.080556
0.139492 0.235601 0.700693 0.453952 2.114852 0.254745 0.063452 0.052500 5.848400 0.303445 0.241094 0.087904 0.189870 5.484236 0.113850 0.628608 0.201094 0.747889
2.924161 0.171995 0.164525 0.315261 0.621323 0.179771 0.465271 0.470140 0.121827 9.533943 1.761439 0.124066 3.038533 0.593478 0.211561 0.408532 1.143980 0.239697 0.165473
"""

_JTT_DCMUT_FREQ = """
0.076862 0.051057 0.042546 0.051269 0.020279 0.041061 0.061820 0.074714 0.022983 0.052569
0.091111 0.059498 0.023414 0.040530 0.050532 0.068225 0.058518 0.014336 0.032303 0.066374
"""

# The same table as the reference's Bio++ 2.x build held it, recovered from the reference's own committed outputs: the
# 190 exchangeabilities and 20 frequencies above, moved (by <= 1.2e-4 relative, i.e. in the sixth decimal the literature
# table is rounded to) so that Myo_unif.vec, Myo_naive.vec and Myo.infos are reproduced to their print precision
# (scripts/fit_jtt_to_fixture.py; the four other fixtures are held out and come out at print precision too,
# tests/test_golden_myoglobin.py).  Used ONLY by the golden tests, to pin oracle and device to the reference at 1e-5 / 2e-6
# instead of 1e-4 / 5e-6; the product default stays the literature table.
_JTT_BPP2X_LOWER = """
0.531677647
0.557966798 0.451094944
0.827444603 0.154899869 5.549530200
0.574478272 1.019842190 0.313310039 0.105626460
0.556724687 3.021994039 0.768833504 0.521645988 0.091310685
1.066680766 0.318482256 0.578115845 7.766556917 0.053902325 3.417705837
1.740159086 1.359652021 0.773313101 1.272434701 0.546391723 0.231293874 1.115631383
0.219970018 3.210669650 4.025778186 1.032341888 0.725003504 5.684079514 0.243768322 0.201695859
0.361684026 0.239195197 0.491003124 0.115968883 0.150558936 0.078268455 0.111772840 0.053768837 0.181788026
0.310006808 0.372261170 0.137288982 0.061485677 0.164593956 0.709003691 0.097484896 0.069491377 0.540570722 2.335139006
0.369436948 6.529254473 2.529516826 0.282465873 0.049007509 2.966732073 1.731683762 0.269839993 0.525096071 0.202561949 0.146481074
0.469395104 0.431044292 0.330718980 0.190000677 0.409202412 0.456900848 0.175083611 0.130379707 0.329659989 4.831666666 3.856906410 0.624580613
0.138292823 0.065314593 0.073480922 0.032522153 0.678335684 0.045682353 0.043829099 0.050212365 0.453428539 0.777090351 2.500294521 0.024520025 0.436180727
1.959599548 0.710489210 0.121803644 0.127163819 0.123653855 1.608125350 0.191994008 0.208081157 1.141961900 0.098579901 1.060504744 0.216344966 0.164220616 0.148481508
3.887095859 1.001551267 5.057963510 0.589268399 2.155331660 0.548807073 0.312449252 1.874296391 0.743455935 0.405119120 0.592511079 0.474477864 0.285564807 0.943970956 2.788406428
4.582565566 0.650281523 2.351311257 0.425158643 0.469822709 0.523825623 0.331584219 0.316862010 0.477356146 2.553806051 0.272513923 0.965640542 2.114727372 0.138903674 1.176961502 4.777648231
0.084325605 1.257961695 0.027701629 0.057463500 1.104176796 0.172207611 0.114394152 0.544178807 0.128194004 0.134510960 0.530328436 0.089135740 0.201311678 0.537922349 0.069965293 0.310924353 0.080552271
0.139494925 0.235601224 0.700693676 0.453951360 2.114851516 0.254742112 0.063451876 0.052500462 5.848399598 0.303445577 0.241094473 0.087907719 0.189875827 5.484237129 0.113841954 0.628608615 0.201096215 0.747887018
2.924162046 0.171995196 0.164525331 0.315261138 0.621320798 0.179773431 0.465271228 0.470139366 0.121826895 9.533943739 1.761438977 0.124065747 3.038533452 0.593478975 0.211562047 0.408531690 1.143980425 0.239697275 0.165468853
"""

_JTT_BPP2X_FREQ = """
0.076861991 0.051057010 0.042546004 0.051268003 0.020278995 0.041061008 0.061820008 0.074713995 0.022983004 0.052568998
0.091110997 0.059498008 0.023414000 0.040529994 0.050531994 0.068224998 0.058517996 0.014336008 0.032303000 0.066374000
"""

# Grantham (1974) chemical distance, upper triangle in AA_ORDER (used by the weighted
# fixtures Myo_*_grantham.vec: nijt=...(weight=AAdist(type=grantham, sym=yes)),
# examples/Proteins/Benchmark/CoMap/analyse.sh:31-43).
_GRANTHAM_UPPER = """
112 111 126 195  91 107  60  86  94  96 106  84 113  27  99  58 148 112  64
     86  96 180  43  54 125  29  97 102  26  91  97 103 110  71 101  77  96
         23 139  46  42  80  68 149 153  94 142 158  91  46  65 174 143 133
            154  61  45  94  81 168 172 101 160 177 108  65  85 181 160 152
                154 170 159 174 198 198 202 196 205 169 112 149 215 194 192
                     29  87  24 109 113  53 101 116  76  68  42 130  99  96
                         98  40 134 138  56 126 140  93  80  65 152 122 121
                             98 135 138 127 127 153  42  56  59 184 147 109
                                 94  99  32  87 100  77  89  47 115  83  84
                                      5 102  10  21  95 142  89  61  33  29
                                        107  15  22  98 145  92  61  36  32
                                             95 102 103 121  78 110  85  97
                                                 28  87 135  81  67  36  21
                                                    114 155 103  40  22  50
                                                         74  38 147 110  68
                                                             58 177 144 124
                                                                128  92  69
                                                                     37  88
                                                                         55
"""


def _lower_to_sym(txt, n=20):
    vals = [float(x) for x in txt.split()]
    assert len(vals) == n * (n - 1) // 2
    S = np.zeros((n, n))
    k = 0
    for i in range(1, n):
        for j in range(i):
            S[i, j] = S[j, i] = vals[k]
            k += 1
    return S


def _upper_to_sym(txt, n=20):
    vals = [float(x) for x in txt.split()]
    assert len(vals) == n * (n - 1) // 2
    S = np.zeros((n, n))
    k = 0
    for i in range(n - 1):
        for j in range(i + 1, n):
            S[i, j] = S[j, i] = vals[k]
            k += 1
    return S


def reversible_generator(exch, freq):
    """Q = S.diag(pi), rows summing to 0, scaled so that -sum_i pi_i Q_ii = 1.

    pi is renormalised to sum to one first (the published JTT frequencies sum to 1.000001).
    Returns (Q row-major [S,S], pi)."""
    pi = np.asarray(freq, dtype=np.float64)
    pi = pi / pi.sum()
    Q = np.asarray(exch, dtype=np.float64) * pi[None, :]
    np.fill_diagonal(Q, 0.0)
    np.fill_diagonal(Q, -Q.sum(axis=1))
    scale = -(pi * np.diag(Q)).sum()
    return Q / scale, pi


def jtt92():
    """JTT92 (DCmut) generator and equilibrium frequencies."""
    S = _lower_to_sym(_JTT_DCMUT_LOWER)
    f = np.array([float(x) for x in _JTT_DCMUT_FREQ.split()])
    return reversible_generator(S, f)


def jtt92_bpp2x():
    """JTT92 as the reference's Bio++ 2.x build held it (fitted to the reference's fixtures, see above): golden tests only."""
    S = _lower_to_sym(_JTT_BPP2X_LOWER)
    f = np.array([float(x) for x in _JTT_BPP2X_FREQ.split()])
    return reversible_generator(S, f)


def grantham_distance():
    return _upper_to_sym(_GRANTHAM_UPPER)


def gtr(a, b, c, d, e, theta, theta1, theta2):
    """Bio++ GTR parameterisation (states A C G T; recalled from bpp-phyl GTR.h, unpinned):
    exchangeabilities S(A,C)=d, S(A,G)=1, S(A,T)=b, S(C,G)=e, S(C,T)=a, S(G,T)=c;
    piA=theta1(1-theta), piC=(1-theta2)theta, piG=theta2 theta, piT=(1-theta1)(1-theta)."""
    pi = np.array([theta1 * (1 - theta), (1 - theta2) * theta, theta2 * theta, (1 - theta1) * (1 - theta)])
    S = np.array([[0, d, 1.0, b], [d, 0, e, a], [1.0, e, 0, c], [b, a, c, 0]], dtype=np.float64)
    return reversible_generator(S, pi)


def synthetic_reversible(nstates, seed):
    """Seeded random reversible generator (throughput work is model-independent)."""
    rng = np.random.default_rng(seed)
    S = rng.gamma(0.7, 1.0, size=(nstates, nstates))
    S = (S + S.T) / 2
    f = rng.dirichlet(np.full(nstates, 5.0))
    return reversible_generator(S, f)


def gamma_rates(alpha, ncat):
    """Equiprobable discrete Gamma(alpha, beta=alpha), class value = mean of the class
    (pinned by max PR in Myo.infos)."""
    from scipy.stats import gamma as _g
    from scipy.special import gammainc
    q = _g.ppf(np.arange(1, ncat) / ncat, a=alpha, scale=1.0 / alpha)
    cuts = np.concatenate([[0.0], q, [np.inf]])
    inc = gammainc(alpha + 1.0, cuts * alpha)
    rates = (inc[1:] - inc[:-1]) * ncat
    probs = np.full(ncat, 1.0 / ncat)
    return rates, probs
