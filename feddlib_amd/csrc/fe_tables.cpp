// Reference-simplex quadrature rules and P1/P2 Lagrange basis tables of the product path.
// These are the constants the assembly kernels stage in LDS.  Behaviour follows
//   FE::getQuadratureValues  feddlib/core/FE/FE_def.hpp:6023-6727
//   FE::phi / FE::gradPhi    feddlib/core/FE/FE_def.hpp:4947-5087 / 5565-5713
//   FE::determineDegree      feddlib/core/FE/FE_def.hpp:5431-5562
#include "fedd_internal.hpp"
#include <cmath>

namespace fedd {

int fe_degree(int nen, int dim, bool grad) {
    const bool p2 = (dim == 2 && nen == 6) || (dim == 3 && nen == 10);
    if (p2) return grad ? 1 : 2;
    return grad ? 0 : 1;
}

static void push(std::vector<double>& pts, std::vector<double>& w, double x, double y, double z,
                 double wt, int dim) {
    pts.push_back(x);
    pts.push_back(y);
    if (dim == 3) pts.push_back(z);
    w.push_back(wt);
}

int fe_quadrature(int dim, int degree, std::vector<double>& pts, std::vector<double>& w) {
    pts.clear();
    w.clear();
    if (degree <= 0) degree = 1;
    if (dim == 2) {
        if (degree == 3 || degree == 4) degree = 5;  // :6070-6071
        if (degree == 1) {
            push(pts, w, 1 / 3., 1 / 3., 0, 1 / 2., 2);
        } else if (degree == 2) {
            const double a = 1 / 6.;
            push(pts, w, 0.5, 0.5, 0, a, 2);
            push(pts, w, 0.0, 0.5, 0, a, 2);
            push(pts, w, 0.5, 0.0, 0, a, 2);
        } else if (degree == 5) {
            const double a = 0.470142064105115, b = 0.101286507323456;
            const double P1 = 0.066197076394253, P2 = 0.062969590272413;
            push(pts, w, 1 / 3., 1 / 3., 0, 9 / 80., 2);
            push(pts, w, a, a, 0, P1, 2);
            push(pts, w, 1 - 2. * a, a, 0, P1, 2);
            push(pts, w, a, 1 - 2. * a, 0, P1, 2);
            push(pts, w, b, b, 0, P2, 2);
            push(pts, w, 1 - 2. * b, b, 0, P2, 2);
            push(pts, w, b, 1 - 2. * b, 0, P2, 2);
        } else {
            set_error("2D quadrature degree %d is not on the hot path", degree);
            return 1;
        }
        return 0;
    }
    if (dim == 3) {
        if (degree == 2) degree = 3;  // :6245-6248
        if (degree == 4) degree = 5;
        if (degree == 1) {
            push(pts, w, .25, .25, .25, 1 / 6., 3);
        } else if (degree == 3) {
            const double a = .25, b = 1. / 6., c = .5, wv = 3. / 40.;
            push(pts, w, a, a, a, -2. / 15., 3);
            push(pts, w, b, b, b, wv, 3);
            push(pts, w, b, b, c, wv, 3);
            push(pts, w, b, c, b, wv, 3);
            push(pts, w, c, b, b, wv, 3);
        } else if (degree == 5) {
            const double s15 = std::sqrt(15.);
            const double a = 0.25, b1 = (7. + s15) / 34., b2 = (7. - s15) / 34.;
            const double c1 = (13. - 3. * s15) / 34., c2 = (13. + 3. * s15) / 34.;
            const double d = (5. - s15) / 20., e = (5. + s15) / 20.;
            const double P1 = (2665. - 14. * s15) / 226800., P2 = (2665. + 14. * s15) / 226800.;
            const double b = 5. / 567.;
            push(pts, w, a, a, a, 8. / 405., 3);
            push(pts, w, b1, b1, b1, P1, 3);
            push(pts, w, b1, b1, c1, P1, 3);
            push(pts, w, b1, c1, b1, P1, 3);
            push(pts, w, c1, b1, b1, P1, 3);
            push(pts, w, b2, b2, b2, P2, 3);
            push(pts, w, b2, b2, c2, P2, 3);
            push(pts, w, b2, c2, b2, P2, 3);
            push(pts, w, c2, b2, b2, P2, 3);
            push(pts, w, d, d, e, b, 3);
            push(pts, w, d, e, d, b, 3);
            push(pts, w, e, d, d, b, 3);
            push(pts, w, d, e, e, b, 3);
            push(pts, w, e, d, e, b, 3);
            push(pts, w, e, e, d, b, 3);
        } else {
            set_error("3D quadrature degree %d is not on the hot path", degree);
            return 1;
        }
        return 0;
    }
    set_error("quadrature: dimension must be 2 or 3");
    return 1;
}

static void basis(int dim, int nen, const double* p, double* ph, double* g) {
    const double x = p[0], y = p[1], z = dim == 3 ? p[2] : 0.0;
    auto G = [&](int i, int d) -> double& { return g[i * dim + d]; };
    for (int i = 0; i < nen * dim; ++i) g[i] = 0.0;
    if (dim == 2 && nen == 3) {
        ph[0] = 1. - x - y; ph[1] = x; ph[2] = y;
        G(0, 0) = -1; G(0, 1) = -1; G(1, 0) = 1; G(2, 1) = 1;
    } else if (dim == 2 && nen == 6) {
        const double l = 1. - x - y;
        ph[0] = -l * (1 - 2. * l); ph[1] = -x * (1 - 2 * x); ph[2] = -y * (1 - 2 * y);
        ph[3] = 4 * x * l; ph[4] = 4 * x * y; ph[5] = 4 * y * l;
        G(0, 0) = 1. - 4. * l; G(0, 1) = 1. - 4. * l;
        G(1, 0) = 4. * x - 1;
        G(2, 1) = 4. * y - 1;
        G(3, 0) = 4 * (1. - 2 * x - y); G(3, 1) = -4 * x;
        G(4, 0) = 4. * y; G(4, 1) = 4. * x;
        G(5, 0) = -4. * y; G(5, 1) = 4 * (1. - x - 2 * y);
    } else if (dim == 3 && nen == 4) {
        ph[0] = 1. - x - y - z; ph[1] = x; ph[2] = y; ph[3] = z;
        G(0, 0) = -1; G(0, 1) = -1; G(0, 2) = -1; G(1, 0) = 1; G(2, 1) = 1; G(3, 2) = 1;
    } else {  // 3D P2, edge order 4=(0,1) 5=(1,2) 6=(0,2) 7=(0,3) 8=(1,3) 9=(2,3)
        const double l = 1. - x - y - z;
        ph[0] = l * (1 - 2 * x - 2 * y - 2 * z); ph[1] = x * (2 * x - 1); ph[2] = y * (2 * y - 1);
        ph[3] = z * (2 * z - 1); ph[4] = 4 * x * l; ph[5] = 4 * x * y; ph[6] = 4 * y * l;
        ph[7] = 4 * z * l; ph[8] = 4 * x * z; ph[9] = 4 * y * z;
        const double s = -3. + 4. * x + 4. * y + 4. * z;
        G(0, 0) = s; G(0, 1) = s; G(0, 2) = s;
        G(1, 0) = 4. * x - 1;
        G(2, 1) = 4. * y - 1;
        G(3, 2) = 4. * z - 1;
        G(4, 0) = 4. - 8. * x - 4. * y - 4. * z; G(4, 1) = -4. * x; G(4, 2) = -4. * x;
        G(5, 0) = 4. * y; G(5, 1) = 4. * x;
        G(6, 0) = -4. * y; G(6, 1) = 4. - 4. * x - 8. * y - 4. * z; G(6, 2) = -4. * y;
        G(7, 0) = -4. * z; G(7, 1) = -4. * z; G(7, 2) = 4. - 4. * x - 4. * y - 8. * z;
        G(8, 0) = 4. * z; G(8, 2) = 4. * x;
        G(9, 1) = 4. * z; G(9, 2) = 4. * y;
    }
}

int fe_tables(int dim, int nen, int degree, FeTables& out) {
    const bool ok = (dim == 2 && (nen == 3 || nen == 6)) || (dim == 3 && (nen == 4 || nen == 10));
    if (!ok) {
        set_error("fe_tables: unsupported element (dim %d, %d nodes)", dim, nen);
        return 1;
    }
    std::vector<double> pts;
    if (fe_quadrature(dim, degree, pts, out.w)) return 1;
    out.dim = dim;
    out.nen = nen;
    out.nq = (int)out.w.size();
    out.phi.assign((size_t)out.nq * nen, 0.0);
    out.dphi.assign((size_t)out.nq * nen * dim, 0.0);
    for (int q = 0; q < out.nq; ++q)
        basis(dim, nen, &pts[(size_t)q * dim], &out.phi[(size_t)q * nen], &out.dphi[(size_t)q * nen * dim]);
    return 0;
}

}  // namespace fedd
