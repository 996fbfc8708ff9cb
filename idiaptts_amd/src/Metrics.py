"""Objective scores used by AcousticModelTrainer.benchmark (reference: idiaptts/src/Metrics.py):
mel-cepstral distortion (nnmnkwii melcd formula 10/ln10 * sqrt(2) * mean_t ||dc||_2, c_0
excluded), F0 RMSE on frames voiced in the original, voicing decision error, band-aperiodicity
distortion. Tiny host-side numpy; needed by the parity harness, not a kernel."""
import math

import numpy as np


class Metrics(object):
    MCD = "MCD"
    F0_RMSE = "F0 RMSE"
    VDE = "VDE"
    BAP_distortion = "BAP distortion"
    Dur_RMSE = "Dur RMSE"
    Dur_pearson = "Dur pearson"

    @staticmethod
    def rmse(org, output, axis=None):
        """sqrt(sum of squared errors / number of ROWS) (reference :165-171: with axis=None the sum
        runs over all elements but the divisor stays the row count)."""
        mse = (org - output) ** 2
        return np.sqrt(mse.sum(axis=axis) / len(mse))

    @staticmethod
    def pearson(org, output):
        """Pearson correlation per column (reference :173-175, scipy.stats.pearsonr)."""
        o = org - org.mean(axis=0)
        p = output - output.mean(axis=0)
        return (o * p).sum(axis=0) / np.sqrt((o * o).sum(axis=0) * (p * p).sum(axis=0))

    @staticmethod
    def melcd(x, y):
        return float(10.0 / np.log(10) * np.sqrt(2.0) * np.sqrt(((x - y) ** 2).sum(-1)).mean())

    @staticmethod
    def mcd_k(org_cep, output_cep, k=None, start_bin=1):
        org = org_cep[:len(output_cep)]
        return Metrics.melcd(output_cep[:, start_bin:k], org[:, start_bin:k])

    @staticmethod
    def f0_rmse(org_lf0, org_vuv, output_lf0):
        org_f0 = np.exp(org_lf0.squeeze())[:len(output_lf0)]
        org_vuv = org_vuv[:len(output_lf0)]
        f0_mse = (org_f0 - np.exp(output_lf0)) ** 2
        return math.sqrt((f0_mse * org_vuv).sum() / org_vuv.sum())

    @staticmethod
    def voicing_decision_error(org_vuv, output_vuv):
        return (org_vuv[:len(output_vuv)] != output_vuv).sum() / len(output_vuv)

    @staticmethod
    def aperiodicity_distortion(org_bap, output_bap):
        org_bap = org_bap[:len(output_bap)]
        if output_bap.ndim > 1 and output_bap.shape[1] > 1:
            return Metrics.mcd_k(org_bap, output_bap)
        return math.sqrt(((org_bap - output_bap) ** 2).mean()) * (10.0 / np.log(10) * np.sqrt(2.0))

    @staticmethod
    def get_metrics(org, output):
        """org / output: (coded_sp, lf0, vuv, bap) tuples -> [MCD, F0 RMSE, VDE, BAP distortion]."""
        o_sp, o_lf0, o_vuv, o_bap = org
        p_sp, p_lf0, p_vuv, p_bap = output
        return [Metrics.mcd_k(o_sp, p_sp), Metrics.f0_rmse(o_lf0, o_vuv, p_lf0),
                Metrics.voicing_decision_error(o_vuv, p_vuv),
                Metrics.aperiodicity_distortion(o_bap, p_bap)]
