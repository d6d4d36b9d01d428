"""Continuous piecewise-linear least squares with given breakpoints.

`learn_dispersion_model` of the reference (modeling/dispersion.pyx:445-467) fits mu(x) and
1/r(x) with the third-party package `pwlf` (Jekel & Venter, "pwlf: a Python library for fitting
1D continuous piecewise linear functions"), which is not a dependency here.  This module is a
restatement of the part of its published method that the reference calls:

    y(x) = b0 + b1 (x - c0) + sum_{i>=1} b_{i+1} (x - c_i) 1[x > c_i]

for breakpoints c0 < c1 < ... ; the coefficients are the linear least-squares solution, optionally
subject to the curve passing through given points (equality constraints, solved through the KKT
system).  `PiecewiseLinFit` offers the same members the reference uses: fit_with_breaks,
fit_with_breaks_force_points, fit_with_breaks_opt (the residual sum of squares as a function of
the interior breakpoints, minimised by the caller), fit_breaks, slopes, intercepts, predict.

Parity note: pwlf is absent from /root/reference and from this image, so these fits are NOT
pinned to pwlf outputs; tests check them against closed-form cases and against numpy.
"""
import numpy as np


class PiecewiseLinFit(object):
    def __init__(self, x, y):
        x, y = np.asarray(x, dtype=np.float64).ravel(), np.asarray(y, dtype=np.float64).ravel()
        if x.size != y.size:
            raise ValueError("x and y differ in length")
        order = np.argsort(x, kind="stable")
        self.x_data, self.y_data = x[order], y[order]
        self.n_data = x.size
        self.break_0, self.break_n = self.x_data[0], self.x_data[-1]
        self.fit_breaks = self.beta = self.slopes = self.intercepts = None
        self.n_segments = self.n_parameters = 0

    # -- design matrix: one column per coefficient of the hinge expansion
    def assemble_regression_matrix(self, breaks, x):
        breaks = np.asarray(breaks, dtype=np.float64)
        x = np.asarray(x, dtype=np.float64)
        cols = [np.ones_like(x), x - breaks[0]]
        for c in breaks[1:-1]:
            cols.append(np.where(x > c, x - c, 0.0))
        return np.column_stack(cols)

    def _set_breaks(self, breaks):
        self.fit_breaks = np.asarray(breaks, dtype=np.float64).copy()
        self.n_segments = self.fit_breaks.size - 1
        self.n_parameters = self.n_segments + 1

    def predict(self, x, beta=None, breaks=None):
        if beta is not None and breaks is not None:
            self.beta = np.asarray(beta, dtype=np.float64)
            self._set_breaks(breaks)
        return self.assemble_regression_matrix(self.fit_breaks, x).dot(self.beta)

    def calc_slopes(self):
        """slope and intercept (value at x = 0 of the extended line) of every segment"""
        at = self.predict(self.fit_breaks)
        self.slopes = np.diff(at) / np.diff(self.fit_breaks)
        self.intercepts = at[:-1] - self.slopes * self.fit_breaks[:-1]
        return self.slopes

    def fit_with_breaks(self, breaks):
        self._set_breaks(breaks)
        A = self.assemble_regression_matrix(self.fit_breaks, self.x_data)
        self.beta, res, _, _ = np.linalg.lstsq(A, self.y_data, rcond=None)
        self.calc_slopes()
        resid = A.dot(self.beta) - self.y_data
        self.ssr = float(resid.dot(resid))
        return self.fit_breaks

    def fit_with_breaks_force_points(self, breaks, x_c, y_c):
        """least squares subject to y(x_c) = y_c: stationarity of the Lagrangian
        [[2 A'A, C'], [C, 0]] [beta; lambda] = [2 A'y; y_c]"""
        self._set_breaks(breaks)
        x_c, y_c = np.atleast_1d(np.asarray(x_c, float)), np.atleast_1d(np.asarray(y_c, float))
        A = self.assemble_regression_matrix(self.fit_breaks, self.x_data)
        Cm = self.assemble_regression_matrix(self.fit_breaks, x_c)
        n, m = self.n_parameters, x_c.size
        K = np.zeros((n + m, n + m))
        K[:n, :n] = 2.0 * A.T.dot(A)
        K[:n, n:] = Cm.T
        K[n:, :n] = Cm
        rhs = np.concatenate([2.0 * A.T.dot(self.y_data), y_c])
        try:
            sol = np.linalg.solve(K, rhs)
        except np.linalg.LinAlgError:
            sol = np.linalg.lstsq(K, rhs, rcond=None)[0]
        self.beta, self.zeta = sol[:n], sol[n:]
        self.calc_slopes()
        resid = A.dot(self.beta) - self.y_data
        self.ssr = float(resid.dot(resid))
        return self.fit_breaks

    def fit_with_breaks_opt(self, var):
        """residual sum of squares of the unconstrained fit whose INTERIOR breakpoints are `var`
        (sorted here; the ends are the data range) -- the objective a caller minimises"""
        inner = np.sort(np.asarray(var, dtype=np.float64))
        breaks = np.concatenate([[self.break_0], inner, [self.break_n]])
        A = self.assemble_regression_matrix(breaks, self.x_data)
        try:
            beta = np.linalg.lstsq(A, self.y_data, rcond=None)[0]
        except np.linalg.LinAlgError:
            return np.inf
        resid = A.dot(beta) - self.y_data
        return float(resid.dot(resid))
