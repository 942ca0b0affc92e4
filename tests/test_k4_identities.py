"""The algebra K4's finish kernel rests on (dsac_amd/csrc/k_backward.hip: k_backward_prep, k_support_scatter), checked on the CPU with the
oracle's Rodrigues stand-in (core/cnn_softam.h:505-509 calls cv::Rodrigues for the 9 x 3 derivative):

  dLoss/drod_i = sum_p C(p)^T (dR/drod_i) X(p) = sum_jk (dR/drod_i)[j][k] M[j][k],      M[j][k] = sum_p C_j X_k      (the VALU form's 9 sums)
               = sum_jm Omega_i[j][m] S[j][m],   Omega_i = (dR/drod_i) R^T,   S[j][m] = sum_p C_j (E - t)_m,  E = R X + t   (the matrix-core form)

and Omega_i is skew-symmetric for a rotation, so only the 6 off-diagonal entries of S are accumulated."""
import numpy as np


def test_omega_is_skew_and_the_off_diagonal_sums_suffice(orc):
    rng = np.random.default_rng(3)
    for trial in range(50):
        rod = rng.normal(size=3) * rng.choice([1e-3, 0.3, 1.0, 2.5])
        R, J = orc.rodrigues_vec2mat(rod, jac=True)  # J: 3 x 9, row i = d R (row-major) / d rod_i
        t = rng.normal(size=3) * 1000
        X = rng.normal(size=(200, 3)) * 1000
        C = rng.normal(size=(200, 3))
        E = X @ R.T + t
        M = C.T @ X                      # M[j][k] = sum_p C_j X_k
        S = C.T @ (E - t)                # S[j][m] = sum_p C_j (E - t)_m
        scale = np.abs(M).max()
        for i in range(3):
            D = J[i].reshape(3, 3)
            Om = D @ R.T
            assert np.abs(Om + Om.T).max() <= 1e-12 * max(1.0, np.abs(Om).max())
            full = (D * M).sum()
            off = sum(Om[j, m] * S[j, m] for j in range(3) for m in range(3) if j != m)
            assert abs(full - off) <= 1e-9 * scale
