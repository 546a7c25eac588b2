"""Hand-built known answers for SURVEY.md Appendix A (the parity-relevant quirks of the reference's control flow), worked
out by hand from the reference's loops and written down as literals -- NOT produced by the oracle or by the kernels,
which are both checked against them (tests/test_quirks_cpu.py, tests/test_gpu_quirks.py).

The case: a put, K = 100, r = 0.05, T = 1, N = 3 steps (decision dates t = 2, 1; the loops stop at t = 1 and never
discount to t = 0), four paths:

    t      0     1     2     3      payoff at t = 1 / 2 / 3
    j0    100    90    95   120          10 /  5 /  0
    j1    100   100    80    70           0 / 20 / 30     (S = K at t = 1: NOT in the money, the test is strict)
    j2    100    95   110    90           5 /  0 / 10
    j3    100    85    85    85          15 / 15 / 15

d = exp(-r T / N) is one step's discount factor."""
import math

import numpy as np

K, R, T, N = 100.0, 0.05, 1.0, 3
D = math.exp(-R * T / N)
S = np.array([[100.0, 100.0, 100.0, 100.0],
              [90.0, 100.0, 95.0, 85.0],
              [95.0, 80.0, 110.0, 85.0],
              [120.0, 70.0, 90.0, 85.0]], np.float64)

# ---- per-step flow (Options_model.py:108-157, options_model_2.py:278-313) with GIVEN continuation values ----------------
# t = 2: cash-flows d * (0, 30, 10, 15); in the money and not exercised: j0 (5), j1 (20), j3 (15).
#        5 > 4 -> j0 exercises; 20 > 20 is false (strict) -> j1 holds; 15 > 16 false -> j3 holds.
# t = 1: cash-flows d * (5, 30 d, 10 d, 15 d); in the money: j0 (10), j2 (5), j3 (15) -- j1 sits on the strike;
#        j0 is already exercised: the mask is STICKY, its larger payoff 10 > 0 is never looked at;
#        5 > 4.9 -> j2 exercises; 15 > 14 -> j3 exercises.
# The loop ends here: values at t = 1, no discount to t = 0.
CONT = np.zeros((N + 1, 4), np.float32)
CONT[2] = [4.0, 20.0, 777.0, 16.0]   # (j2 is out of the money at t = 2: its entry is never read)
CONT[1] = [0.0, 0.0, 4.9, 14.0]      # (j0: exercised; j1: not in the money -- never read)
PER_STEP_CF = np.array([5.0 * D, 30.0 * D * D, 5.0, 15.0])
PER_STEP_EX = np.array([True, False, True, True])
PER_STEP_TEX = np.array([2, 3, 1, 1])      # exercise date (N = held to expiry)
PER_STEP_NITM = {2: 3, 1: 2}               # regression-set sizes: in the money AND not yet exercised
# textbook Longstaff-Schwartz on the same values: no sticky mask -> j0 exercises again at t = 1 (10 > 0 overwrites 5 d),
# and the result is discounted to t = 0
TEXTBOOK_CF = np.array([10.0 * D, 30.0 * D * D * D, 5.0 * D, 15.0 * D])

# ---- two-pass flow (options_model_3.py:482-516, 615-651) ------------------------------------------------------------
# Pass 1 takes NO decisions: the regression target of every in-the-money (t, path) is the discounted TERMINAL payoff.
#   t = 2: j0 (x = 95, y = 0), j1 (80, 30 d), j3 (85, 15 d);   t = 1: j0 (90, 0), j2 (95, 10 d^2), j3 (85, 15 d^2)
# (j1 at t = 1 sits on the strike: no row; j2 at t = 2 is out of the money).  Rows in the reference's order: t
# descending, paths ascending.
ROWS_T = np.array([2, 2, 2, 1, 1, 1])
ROWS_X = np.array([95.0, 80.0, 85.0, 90.0, 95.0, 85.0])
ROWS_Y = np.array([0.0, 30.0 * D, 15.0 * D, 0.0, 10.0 * D * D, 15.0 * D * D])
# Pass 2 with a frozen continuation function that is CONSTANT per step (c2 = 15 at t = 2, c1 = 9.99 at t = 1):
#   t = 2: j0 5 > 15 no; j1 20 > 15 yes; j3 15 > 15 NO (strict).   t = 1: j0 10 > 9.99 yes; j1 exercised (and on the
#   strike); j2 5 > 9.99 no; j3 15 > 9.99 yes.   Values at t = 1.
FROZEN_C = {2: 15.0, 1: 9.99}
TWO_PASS_CF = np.array([10.0, 20.0 * D, 10.0 * D * D, 15.0])
TWO_PASS_EX = np.array([True, True, False, True])
TWO_PASS_TEX = np.array([1, 2, 3, 1])


def features(spot, t_current):
    """options_model_3.py:105-121 by hand: x = S / K, s = sqrt(max(T - t, 1e-6)); [1, x, x^2, x^3, max(x - 1, 0), s, x s]
    (the function's `r` argument is never used)."""
    x = spot / K
    s = math.sqrt(max(T - t_current, 1e-6))
    return [1.0, x, x * x, x * x * x, max(x - 1.0, 0.0), s, x * s]
