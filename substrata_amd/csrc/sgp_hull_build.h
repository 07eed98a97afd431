// sgp_hull_build.h -- HOST side of the convex hull shapes: hull from a point cloud, volume / centre of mass / inertia, body frame.
//
// Role of JPH::ConvexHullShapeSettings::Create + MassProperties (+ OffsetCenterOfMassShape) (/root/reference/gui_client/
// CarPhysics.cpp:66-92, BikePhysics.cpp:76-112): brute-force supporting planes for <= 32 points (rounds 1-4, unchanged), an incremental hull with
// one face per supporting plane for 33 .. 256 (round 5; 256 = JPH::ConvexHullShape::cMaxPointsInHull), signed-tetrahedra mass properties, Jacobi
// principal axes; double precision, rounded to float once.
#pragma once
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "sgp_device_collide.h"     // sgd_hull

static inline v3 sgh_v3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }

struct sgh_d3 { double x, y, z; };
static inline sgh_d3 sgh_d3_sub(sgh_d3 a, sgh_d3 b) { sgh_d3 r = { a.x - b.x, a.y - b.y, a.z - b.z }; return r; }
static inline sgh_d3 sgh_d3_cross(sgh_d3 a, sgh_d3 b) { sgh_d3 r = { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; return r; }
static inline double sgh_d3_dot(sgh_d3 a, sgh_d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// Jacobi eigen-decomposition of a symmetric 3x3 matrix: a -> diagonal, v = eigenvectors (columns).
static inline void sgh_jacobi3(double a[3][3], double v[3][3])
{
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) v[i][j] = i == j ? 1.0 : 0.0;
	for (int sweep = 0; sweep < 64; ++sweep) {
		const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
		if (off < 1.0e-14 * (fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]) + 1.0e-300)) break;
		for (int p = 0; p < 2; ++p) for (int q = p + 1; q < 3; ++q) {
			if (fabs(a[p][q]) < 1.0e-300) continue;
			const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
			const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
			const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
			for (int k = 0; k < 3; ++k) { const double akp = a[k][p], akq = a[k][q]; a[k][p] = c * akp - s * akq; a[k][q] = s * akp + c * akq; }
			for (int k = 0; k < 3; ++k) { const double apk = a[p][k], aqk = a[q][k]; a[p][k] = c * apk - s * aqk; a[q][k] = s * apk + c * aqk; }
			for (int k = 0; k < 3; ++k) { const double vkp = v[k][p], vkq = v[k][q]; v[k][p] = c * vkp - s * vkq; v[k][q] = s * vkp + c * vkq; }
		}
	}
}

/* A candidate for a face's plane: three points and twice the area of their triangle. */
typedef struct { double len; unsigned char a, b, c; } sgh_cand;
static inline int sgh_cand_cmp(const void* x, const void* y)
{
	const sgh_cand* p = (const sgh_cand*)x; const sgh_cand* q = (const sgh_cand*)y;
	if (p->len != q->len) return p->len > q->len ? -1 : 1;      /* largest first */
	if (p->a != q->a) return (int)p->a - (int)q->a;
	if (p->b != q->b) return (int)p->b - (int)q->b;
	return (int)p->c - (int)q->c;
}

/* Faces from candidate triangles (round 6): one face per supporting plane = ALL points of the cloud within eps of it, reduced to the corners of
   their polygon.  The candidates are walked from the largest triangle down, so a face takes its plane from a well-conditioned triangle -- on a cap whose points
   are coplanar only to float rounding the first triangle met can be a sliver with a garbage normal (round 5 took that one) --, a candidate whose corners already
   lie in an accepted face is that face again, and a candidate whose plane has points on both sides (a triangle of a hull that went wrong) is no face.
   Returns the number of faces (planes fn / fd, corners per face in fmem / fmem_start, ascending point index), < 0 when the tables overflow. */
static inline int sgh_faces_from_candidates(const sgh_d3* pts, int n, double eps, double ext, sgh_cand* cand, int ncand, sgh_d3* fn, double* fd, unsigned short* fmem_start, unsigned char* fmem, int fmem_cap)
{
	int nf = 0, nm = 0;
	qsort(cand, (size_t)ncand, sizeof(sgh_cand), sgh_cand_cmp);
	for (int ci = 0; ci < ncand; ++ci) {
		if (!(cand[ci].len > 1.0e-9 * ext * ext)) break;           /* (sorted: the rest is degenerate) */
		const sgh_d3 pa = pts[cand[ci].a], pb = pts[cand[ci].b], pc = pts[cand[ci].c];
		int known = 0;
		for (int f = 0; f < nf && !known; ++f)
			if (fabs(sgh_d3_dot(fn[f], pa) - fd[f]) <= eps && fabs(sgh_d3_dot(fn[f], pb) - fd[f]) <= eps && fabs(sgh_d3_dot(fn[f], pc) - fd[f]) <= eps) known = 1;
		if (known) continue;
		sgh_d3 nn = sgh_d3_cross(sgh_d3_sub(pb, pa), sgh_d3_sub(pc, pa));
		const double len = sqrt(sgh_d3_dot(nn, nn));
		if (!(len > 0.0)) continue;
		nn.x /= len; nn.y /= len; nn.z /= len;
		const double d0 = sgh_d3_dot(nn, pa);
		double mx = 0.0, mn = 0.0;
		for (int q = 0; q < n; ++q) { const double sd = sgh_d3_dot(nn, pts[q]) - d0; if (sd > mx) mx = sd; if (sd < mn) mn = sd; }
		if (mx > eps && mn < -eps) continue;                       /* points on both sides: not a supporting plane */
		if (mx > eps) { nn.x = -nn.x; nn.y = -nn.y; nn.z = -nn.z; }
		const double dd = sgh_d3_dot(nn, pa);
		if (nf == SGD_HULL_MAX_FACES) return -2;
		fn[nf] = nn; fd[nf] = dd; fmem_start[nf] = (unsigned short)nm;
		int cnt = 0;
		for (int q = 0; q < n; ++q) if (fabs(sgh_d3_dot(nn, pts[q]) - dd) <= eps) { if (nm == fmem_cap) return -2; fmem[nm++] = (unsigned char)q; ++cnt; }
		/* only the corners of the face's polygon stay: points inside a face or along one of its sides (a tessellated flat side) are not vertices of the
		   solid.  Convex hull of the members in the face's plane (monotone chain). */
		{
			const int m0 = fmem_start[nf];
			sgh_d3 u = sgh_d3_sub(pts[fmem[m0 + 1]], pts[fmem[m0]]);
			for (int k = m0 + 2; k < nm; ++k) { const sgh_d3 u2 = sgh_d3_sub(pts[fmem[k]], pts[fmem[m0]]); if (sgh_d3_dot(u2, u2) > sgh_d3_dot(u, u)) u = u2; }
			const double ul = sqrt(sgh_d3_dot(u, u)); u.x /= ul; u.y /= ul; u.z /= ul;
			const sgh_d3 w = sgh_d3_cross(nn, u);
			double px[256], py[256]; int ord[256];
			for (int k = 0; k < cnt; ++k) { const sgh_d3 r = sgh_d3_sub(pts[fmem[m0 + k]], pts[fmem[m0]]); px[k] = sgh_d3_dot(r, u); py[k] = sgh_d3_dot(r, w); ord[k] = k; }
			for (int a = 1; a < cnt; ++a) { const int o = ord[a]; int bb = a - 1; while (bb >= 0 && (px[ord[bb]] > px[o] || (px[ord[bb]] == px[o] && py[ord[bb]] > py[o]))) { ord[bb + 1] = ord[bb]; --bb; } ord[bb + 1] = o; }
			int hullk[512]; int hk = 0;
			const double tol = eps * ul;                       /* (twice the area of a triangle as thin as eps over the face's extent) */
			for (int pass = 0; pass < 2; ++pass) {
				const int base = hk;
				for (int ii = 0; ii < cnt; ++ii) {
					const int o = pass == 0 ? ord[ii] : ord[cnt - 1 - ii];
					while (hk - base >= 2) {
						const int o1 = hullk[hk - 1], o0 = hullk[hk - 2];
						const double cr = (px[o1] - px[o0]) * (py[o] - py[o0]) - (py[o1] - py[o0]) * (px[o] - px[o0]);
						if (cr <= tol) --hk; else break;
					}
					hullk[hk++] = o;
				}
				--hk;                                          /* (the last point of a chain is the first of the next) */
			}
			unsigned char keepm[256]; memset(keepm, 0, sizeof(keepm));
			for (int k = 0; k < hk; ++k) keepm[hullk[k]] = 1;
			int m1 = m0;
			for (int k = 0; k < cnt; ++k) if (keepm[k]) fmem[m1++] = fmem[m0 + k];
			nm = m1; cnt = m1 - m0;
		}
		if (cnt < 3) { nm = fmem_start[nf]; continue; }
		++nf;
	}
	fmem_start[nf] = (unsigned short)nm;
	/* A face whose corners all belong to another face is that face seen from a plane a hair off.  It goes.  Two passes: which faces are covered is
	   decided against the untouched arrays, then the arrays are compacted. */
	{
		unsigned char covered[SGD_HULL_MAX_FACES + 1]; memset(covered, 0, sizeof(covered));
		for (int f = 0; f < nf; ++f) {
			const int f0 = fmem_start[f], cf = fmem_start[f + 1] - f0;
			for (int g = 0; g < nf && !covered[f]; ++g) {
				if (g == f) continue;
				const int g0 = fmem_start[g], cg = fmem_start[g + 1] - g0;
				if (cg < cf || (cg == cf && g > f)) continue;
				int all = 1;
				for (int k = 0; k < cf && all; ++k) { int found = 0; for (int m = 0; m < cg; ++m) if (fmem[g0 + m] == fmem[f0 + k]) { found = 1; break; } all = found; }
				covered[f] = (unsigned char)all;
			}
		}
		int nf2 = 0, nm2 = 0;
		for (int f = 0; f < nf; ++f) {
			if (covered[f]) continue;
			const int f0 = fmem_start[f], f1 = fmem_start[f + 1];      /* (read before entry nf2 <= f is overwritten; entry f + 1 is still the old one) */
			fn[nf2] = fn[f]; fd[nf2] = fd[f];
			for (int k = f0; k < f1; ++k) fmem[nm2 + (k - f0)] = fmem[k];      /* (nm2 <= f0: moving down, never over unread members) */
			fmem_start[nf2] = (unsigned short)nm2; nm2 += f1 - f0; ++nf2;
		}
		fmem_start[nf2] = (unsigned short)nm2; nf = nf2;
	}
	return nf;
}

/* Faces of a cloud of 33 .. 256 points (round 5): incremental hull over the points in index order (initial tetrahedron from the extreme points, then every
   point outside the hull so far replaces the triangles it sees by a fan from their horizon); its triangles are the candidates sgh_faces_from_candidates
   assembles the faces from (round 6).  Deterministic: ties go to the lower index everywhere.  Returns the number of faces, < 0 on a degenerate cloud. */
static inline int sgh_hull_faces_large(const sgh_d3* pts, int n, double eps, double ext, sgh_d3* fn, double* fd, unsigned short* fmem_start, unsigned char* fmem, int fmem_cap)
{
	enum { TCAP = 8192 };
	typedef struct { int a, b, c, alive; sgh_d3 n; double d, len; } tri_t;      /* len: twice the area */
	tri_t* T = (tri_t*)malloc(sizeof(tri_t) * TCAP);
	short* etri = (short*)malloc(sizeof(short) * 256 * 256);      // triangle holding the directed edge a -> b
	int nt = 0, result = -1;
	unsigned char used[256]; memset(used, 0, sizeof(used));
	if (!T || !etri) goto done;
	memset(etri, 0xFF, sizeof(short) * 256 * 256);      // -1: no triangle has held this directed edge (round 6: the table was read uninitialised where a visible
	                                                       // triangle's edge had no twin -- a hull gone non-manifold within eps -- and indexed T with whatever malloc left)
	{
		// initial tetrahedron
		int p0 = 0, p1 = -1, p2 = -1, p3 = -1; double best;
		for (int i = 1; i < n; ++i) if (pts[i].x < pts[p0].x) p0 = i;
		best = 0.0; for (int i = 0; i < n; ++i) { const sgh_d3 d = sgh_d3_sub(pts[i], pts[p0]); const double l = sgh_d3_dot(d, d); if (l > best) { best = l; p1 = i; } }
		if (p1 < 0) goto done;
		best = 0.0; for (int i = 0; i < n; ++i) { const sgh_d3 c = sgh_d3_cross(sgh_d3_sub(pts[p1], pts[p0]), sgh_d3_sub(pts[i], pts[p0])); const double l = sgh_d3_dot(c, c); if (l > best) { best = l; p2 = i; } }
		if (p2 < 0 || best < 1.0e-18 * ext * ext * ext * ext) goto done;
		{
			sgh_d3 nn = sgh_d3_cross(sgh_d3_sub(pts[p1], pts[p0]), sgh_d3_sub(pts[p2], pts[p0]));
			const double len = sqrt(sgh_d3_dot(nn, nn)); nn.x /= len; nn.y /= len; nn.z /= len;
			best = 0.0; for (int i = 0; i < n; ++i) { const double s = fabs(sgh_d3_dot(nn, sgh_d3_sub(pts[i], pts[p0]))); if (s > best) { best = s; p3 = i; } }
			if (p3 < 0 || best <= eps) goto done;                    // flat cloud
		}
		const sgh_d3 cen = { (pts[p0].x + pts[p1].x + pts[p2].x + pts[p3].x) / 4.0, (pts[p0].y + pts[p1].y + pts[p2].y + pts[p3].y) / 4.0, (pts[p0].z + pts[p1].z + pts[p2].z + pts[p3].z) / 4.0 };
		const int tet[4][3] = { { p0, p1, p2 }, { p0, p1, p3 }, { p0, p2, p3 }, { p1, p2, p3 } };
		for (int t = 0; t < 4; ++t) {
			tri_t tr; tr.a = tet[t][0]; tr.b = tet[t][1]; tr.c = tet[t][2]; tr.alive = 1;
			sgh_d3 nn = sgh_d3_cross(sgh_d3_sub(pts[tr.b], pts[tr.a]), sgh_d3_sub(pts[tr.c], pts[tr.a]));
			double len = sqrt(sgh_d3_dot(nn, nn)); nn.x /= len; nn.y /= len; nn.z /= len;
			if (sgh_d3_dot(nn, cen) - sgh_d3_dot(nn, pts[tr.a]) > 0.0) { const int tmp = tr.b; tr.b = tr.c; tr.c = tmp; nn.x = -nn.x; nn.y = -nn.y; nn.z = -nn.z; }
			tr.n = nn; tr.d = sgh_d3_dot(nn, pts[tr.a]); tr.len = len;
			T[nt++] = tr;
		}
		used[p0] = used[p1] = used[p2] = used[p3] = 1;
		// the other points in index order
		for (int q = 0; q < n; ++q) {
			if (used[q]) continue;
			int any = 0;
			for (int t = 0; t < nt; ++t) if (T[t].alive && sgh_d3_dot(T[t].n, pts[q]) - T[t].d > eps) { any = 1; break; }
			if (!any) continue;                                    // inside (or on) the hull so far
			for (int t = 0; t < nt; ++t) if (T[t].alive) { etri[T[t].a * 256 + T[t].b] = (short)t; etri[T[t].b * 256 + T[t].c] = (short)t; etri[T[t].c * 256 + T[t].a] = (short)t; }
			const int nt0 = nt;
			// visible triangles; horizon = their edges whose twin belongs to a triangle that is not visible
			for (int t = 0; t < nt0; ++t) if (T[t].alive && sgh_d3_dot(T[t].n, pts[q]) - T[t].d > eps) T[t].alive = 2;
			for (int t = 0; t < nt0; ++t) {
				if (T[t].alive != 2) continue;
				const int e[3][2] = { { T[t].a, T[t].b }, { T[t].b, T[t].c }, { T[t].c, T[t].a } };
				for (int k = 0; k < 3; ++k) {
					const int tw = etri[e[k][1] * 256 + e[k][0]];
					if (tw >= 0 && T[tw].alive == 2) continue;      // (no twin, or a dead one: the edge is on the horizon)
					if (nt == TCAP) goto done;
					tri_t tr; tr.a = e[k][0]; tr.b = e[k][1]; tr.c = q; tr.alive = 1;
					sgh_d3 nn = sgh_d3_cross(sgh_d3_sub(pts[tr.b], pts[tr.a]), sgh_d3_sub(pts[tr.c], pts[tr.a]));
					const double len = sqrt(sgh_d3_dot(nn, nn));
					if (!(len > 0.0)) goto done;
					nn.x /= len; nn.y /= len; nn.z /= len;
					tr.n = nn; tr.d = sgh_d3_dot(nn, pts[tr.a]); tr.len = len;
					T[nt++] = tr;
				}
			}
			for (int t = 0; t < nt0; ++t) if (T[t].alive == 2) T[t].alive = 0;
			used[q] = 1;
		}
		/* the triangles that are left are the candidates for the faces' planes */
		{
			sgh_cand* cand = (sgh_cand*)malloc(sizeof(sgh_cand) * (size_t)(nt + 1));
			if (!cand) goto done;
			int nc = 0;
			for (int t = 0; t < nt; ++t) if (T[t].alive) { cand[nc].len = T[t].len; cand[nc].a = (unsigned char)T[t].a; cand[nc].b = (unsigned char)T[t].b; cand[nc].c = (unsigned char)T[t].c; ++nc; }
			result = sgh_faces_from_candidates(pts, n, eps, ext, cand, nc, fn, fd, fmem_start, fmem, fmem_cap);
			free(cand);
		}
	}
done:
	free(T); free(etri);
	return result;
}

/* Steps 2 - 7 of the builder on the unique points.  mode 0: <= 32 points by the brute-force supporting-plane search of rounds 1-4 (bit-identical for every cloud
   whose coplanar points are coplanar exactly: boxes, the car hull, every fixture), more through the incremental hull; mode 1 (<= 32 points): every point
   triple is a candidate for sgh_faces_from_candidates.  Returns 0, or < 0: -3 = the result failed the checks a hull must pass (round 6) -- every
   face plane supports the cloud, every edge runs between exactly two faces, V - E + F = 2.  With vertices from the cloud these three make the result THE hull,
   whatever path the faces came by. */
static inline int sgd_hull_build_pts(const sgh_d3* pts, int n, double eps, double ext, int mode, const float* com_offset, sgd_hull* h, float com_out[3], float rot_out[4])
{
	memset(h, 0, sizeof(*h));
	// 2. faces: planes fn / fd, and per face the points on it (ascending index)
	static const int FCAP = SGD_HULL_MAX_FACES + 64;
	sgh_d3* fn = (sgh_d3*)malloc(sizeof(sgh_d3) * FCAP); double* fd = (double*)malloc(sizeof(double) * FCAP);
	unsigned short* fmem_start = (unsigned short*)malloc(sizeof(unsigned short) * (FCAP + 1)); unsigned char* fmem = (unsigned char*)malloc(8192);
	int nf = 0, rc = 0;
#define SGH_FAIL(code) do { rc = (code); goto cleanup; } while (0)
	if (n <= SGD_HULL_SMALL_VERTS && mode == 1) {
		sgh_cand* cand = (sgh_cand*)malloc(sizeof(sgh_cand) * 4960);      /* 32 choose 3 */
		if (!cand) SGH_FAIL(-1);
		int nc = 0;
		for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) for (int k = j + 1; k < n; ++k) {
			const sgh_d3 cr = sgh_d3_cross(sgh_d3_sub(pts[j], pts[i]), sgh_d3_sub(pts[k], pts[i]));
			cand[nc].len = sqrt(sgh_d3_dot(cr, cr)); cand[nc].a = (unsigned char)i; cand[nc].b = (unsigned char)j; cand[nc].c = (unsigned char)k; ++nc;
		}
		nf = sgh_faces_from_candidates(pts, n, eps, ext, cand, nc, fn, fd, fmem_start, fmem, 8192);
		free(cand);
		if (nf < 0) SGH_FAIL(nf);
	} else if (n <= SGD_HULL_SMALL_VERTS) {
		// up to 32 points (rounds 1-4, unchanged): every supporting plane through three points; face = all points on that plane
		unsigned int masks[64]; int nm = 0;
		for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) for (int k = j + 1; k < n; ++k) {
			sgh_d3 nn = sgh_d3_cross(sgh_d3_sub(pts[j], pts[i]), sgh_d3_sub(pts[k], pts[i]));
			const double len = sqrt(sgh_d3_dot(nn, nn));
			if (len < 1.0e-9 * ext * ext) continue;
			nn.x /= len; nn.y /= len; nn.z /= len;
			const double d0 = sgh_d3_dot(nn, pts[i]);
			double mx = 0.0, mn = 0.0;
			for (int q = 0; q < n; ++q) { const double s = sgh_d3_dot(nn, pts[q]) - d0; if (s > mx) mx = s; if (s < mn) mn = s; }
			if (mx > eps && mn < -eps) continue;                 // points on both sides: not a supporting plane
			if (mx > eps) { nn.x = -nn.x; nn.y = -nn.y; nn.z = -nn.z; }
			const double dd = sgh_d3_dot(nn, pts[i]);
			unsigned int mask = 0; int cnt = 0;
			for (int q = 0; q < n; ++q) if (fabs(sgh_d3_dot(nn, pts[q]) - dd) <= eps) { mask |= 1u << q; ++cnt; }
			if (cnt < 3) continue;
			int known = 0;
			for (int f = 0; f < nf; ++f) if (masks[f] == mask) { known = 1; break; }
			if (known) continue;
			if (nf == 60) SGH_FAIL(-2);      // (the limit of rounds 1-4 on this path; a face may hold every point since round 5)
			masks[nf] = mask; fn[nf] = nn; fd[nf] = dd; fmem_start[nf] = (unsigned short)nm;
			for (int q = 0; q < n; ++q) if (mask & (1u << q)) fmem[nm++] = (unsigned char)q;
			++nf;
		}
		fmem_start[nf] = (unsigned short)nm;
	} else {
		nf = sgh_hull_faces_large(pts, n, eps, ext, fn, fd, fmem_start, fmem, 8192);
		if (nf < 0) SGH_FAIL(nf);
	}
	if (nf < 4) SGH_FAIL(-1);                             // flat or degenerate cloud
	/* check 1 of 3: every face plane supports the cloud (no point beyond it) */
	for (int f = 0; f < nf; ++f) for (int q = 0; q < n; ++q) if (sgh_d3_dot(fn[f], pts[q]) - fd[f] > 2.0 * eps) SGH_FAIL(-3);
	{
	/* 3. drop interior points, order each face counter-clockwise seen from outside.  A face keeps all its corners (a 64-gon cap is ONE face, one SAT axis,
	      one supporting face); the contact manifold clips against every (cnt + 15) / 16-th of them, sgd_hull_face_contact -- as ConvexHullShape::GetSupportingFace
	      thins a face that would overflow its caller's buffer.  UNVERIFIED: upstream. */
	unsigned char usedv[256]; memset(usedv, 0, sizeof(usedv));
	for (int k = 0; k < fmem_start[nf]; ++k) usedv[fmem[k]] = 1;
	int remap[256]; int nv = 0;
	sgh_d3 hv[SGD_HULL_MAX_VERTS];
	for (int q = 0; q < n; ++q) { if (usedv[q]) { remap[q] = nv; hv[nv++] = pts[q]; } else remap[q] = -1; }
	int* fstart_ = (int*)malloc(sizeof(int) * (SGD_HULL_MAX_FACES + 1)); int* fidx_ = (int*)malloc(sizeof(int) * SGD_HULL_MAX_FACE_IDX);
	sgh_d3* fn2 = (sgh_d3*)malloc(sizeof(sgh_d3) * SGD_HULL_MAX_FACES); double* fd2 = (double*)malloc(sizeof(double) * SGD_HULL_MAX_FACES);
	int nidx = 0, nf2 = 0;
	for (int f = 0; f < nf && rc == 0; ++f) {
		int ids[256]; double ang[256]; int cnt = 0;
		sgh_d3 c = { 0, 0, 0 };
		for (int k = fmem_start[f]; k < fmem_start[f + 1]; ++k) { const int q = fmem[k]; ids[cnt++] = remap[q]; c.x += pts[q].x; c.y += pts[q].y; c.z += pts[q].z; }
		c.x /= cnt; c.y /= cnt; c.z /= cnt;
		sgh_d3 u = sgh_d3_sub(hv[ids[0]], c);
		const double ul = sqrt(sgh_d3_dot(u, u)); u.x /= ul; u.y /= ul; u.z /= ul;
		const sgh_d3 w = sgh_d3_cross(fn[f], u);
		for (int k = 0; k < cnt; ++k) { const sgh_d3 r = sgh_d3_sub(hv[ids[k]], c); ang[k] = atan2(sgh_d3_dot(r, w), sgh_d3_dot(r, u)); }
		for (int a = 1; a < cnt; ++a) { const int id = ids[a]; const double av = ang[a]; int b = a - 1; while (b >= 0 && ang[b] > av) { ids[b + 1] = ids[b]; ang[b + 1] = ang[b]; --b; } ids[b + 1] = id; ang[b + 1] = av; }
		if (nf2 == SGD_HULL_MAX_FACES || nidx + cnt > SGD_HULL_MAX_FACE_IDX) { rc = -2; break; }
		fstart_[nf2] = nidx; fn2[nf2] = fn[f]; fd2[nf2] = fd[f]; ++nf2;
		for (int k = 0; k < cnt; ++k) fidx_[nidx++] = ids[k];
	}
	fstart_[nf2] = nidx;
	if (rc == 0) {
	nf = nf2;
	int* const fstart = fstart_; int* const fidx = fidx_;
	sgh_d3* const fn = fn2; double* const fd = fd2;
	// 4. volume, centre of mass, inertia about the origin (signed tetrahedra of the fan-triangulated faces)
	double vol = 0.0; sgh_d3 cm = { 0, 0, 0 };
	double xx = 0, yy = 0, zz = 0, xy = 0, xz = 0, yz = 0;
	for (int f = 0; f < nf; ++f) {
		for (int k = fstart[f] + 1; k + 1 < fstart[f + 1]; ++k) {
			const sgh_d3 a = hv[fidx[fstart[f]]], b = hv[fidx[k]], c = hv[fidx[k + 1]];
			const double det = sgh_d3_dot(a, sgh_d3_cross(b, c));
			vol += det / 6.0;
			cm.x += det / 24.0 * (a.x + b.x + c.x); cm.y += det / 24.0 * (a.y + b.y + c.y); cm.z += det / 24.0 * (a.z + b.z + c.z);
			xx += det / 60.0 * (a.x * a.x + b.x * b.x + c.x * c.x + a.x * b.x + a.x * c.x + b.x * c.x);
			yy += det / 60.0 * (a.y * a.y + b.y * b.y + c.y * c.y + a.y * b.y + a.y * c.y + b.y * c.y);
			zz += det / 60.0 * (a.z * a.z + b.z * b.z + c.z * c.z + a.z * b.z + a.z * c.z + b.z * c.z);
			xy += det / 120.0 * (2 * a.x * a.y + 2 * b.x * b.y + 2 * c.x * c.y + a.x * b.y + a.y * b.x + a.x * c.y + a.y * c.x + b.x * c.y + b.y * c.x);
			xz += det / 120.0 * (2 * a.x * a.z + 2 * b.x * b.z + 2 * c.x * c.z + a.x * b.z + a.z * b.x + a.x * c.z + a.z * c.x + b.x * c.z + b.z * c.x);
			yz += det / 120.0 * (2 * a.y * a.z + 2 * b.y * b.z + 2 * c.y * c.z + a.y * b.z + a.z * b.y + a.y * c.z + a.z * c.y + b.y * c.z + b.z * c.y);
		}
	}
	if (!(vol > 1.0e-12 * ext * ext * ext)) { rc = -1; goto finish; }
	cm.x /= vol; cm.y /= vol; cm.z /= vol;
	// second moments about the centre of mass (of the hull's own mass distribution)
	xx -= vol * cm.x * cm.x; yy -= vol * cm.y * cm.y; zz -= vol * cm.z * cm.z;
	xy -= vol * cm.x * cm.y; xz -= vol * cm.x * cm.z; yz -= vol * cm.y * cm.z;
	if (com_offset) {
		// the body origin moves to cm + o: second moments about it gain vol * o o^T (parallel axis theorem)
		const sgh_d3 o = { com_offset[0], com_offset[1], com_offset[2] };
		xx += vol * o.x * o.x; yy += vol * o.y * o.y; zz += vol * o.z * o.z; xy += vol * o.x * o.y; xz += vol * o.x * o.z; yz += vol * o.y * o.z;
		cm.x += o.x; cm.y += o.y; cm.z += o.z;
	}
	double I[3][3] = { { yy + zz, -xy, -xz }, { -xy, xx + zz, -yz }, { -xz, -yz, xx + yy } }, V[3][3];
	sgh_jacobi3(I, V);
	// right-handed frame
	{
		const sgh_d3 c0 = { V[0][0], V[1][0], V[2][0] }, c1 = { V[0][1], V[1][1], V[2][1] }, c2 = { V[0][2], V[1][2], V[2][2] };
		if (sgh_d3_dot(sgh_d3_cross(c0, c1), c2) < 0.0) { V[0][2] = -V[0][2]; V[1][2] = -V[1][2]; V[2][2] = -V[2][2]; }
	}
	// 6. into the body frame
	h->nv = nv; h->nf = nf;
	float bmin[3] = { 3.4e38f, 3.4e38f, 3.4e38f }, bmax[3] = { -3.4e38f, -3.4e38f, -3.4e38f }; float br = 0.0f;
	for (int i = 0; i < nv; ++i) {
		const sgh_d3 r = sgh_d3_sub(hv[i], cm);
		const double lx = V[0][0] * r.x + V[1][0] * r.y + V[2][0] * r.z, ly = V[0][1] * r.x + V[1][1] * r.y + V[2][1] * r.z, lz = V[0][2] * r.x + V[1][2] * r.y + V[2][2] * r.z;
		h->verts[i] = sgh_v3((float)lx, (float)ly, (float)lz);
		const float c[3] = { (float)lx, (float)ly, (float)lz };
		for (int k = 0; k < 3; ++k) { if (c[k] < bmin[k]) bmin[k] = c[k]; if (c[k] > bmax[k]) bmax[k] = c[k]; }
		const float rl = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]); if (rl > br) br = rl;
	}
	for (int f = 0; f < nf; ++f) {
		const sgh_d3 nn = fn[f];
		h->normals[f] = sgh_v3((float)(V[0][0] * nn.x + V[1][0] * nn.y + V[2][0] * nn.z), (float)(V[0][1] * nn.x + V[1][1] * nn.y + V[2][1] * nn.z), (float)(V[0][2] * nn.x + V[1][2] * nn.y + V[2][2] * nn.z));
		h->plane_d[f] = (float)(fd[f] - sgh_d3_dot(nn, cm));
		h->face_start[f] = (unsigned short)fstart[f];
	}
	h->face_start[nf] = (unsigned short)fstart[nf];
	for (int k = 0; k < nidx; ++k) h->face_idx[k] = (unsigned char)fidx[k];
	// 7. edges: every consecutive pair of a face loop, once
	int ne = 0;
	for (int f = 0; f < nf; ++f) for (int k = fstart[f]; k < fstart[f + 1]; ++k) {
		const int a = fidx[k], b = fidx[k + 1 < fstart[f + 1] ? k + 1 : fstart[f]];
		const int lo = a < b ? a : b, hi = a < b ? b : a;
		int known = 0;
		for (int e = 0; e < ne; ++e) if (h->edge_a[e] == lo && h->edge_b[e] == hi) { known = 1; break; }
		if (known) continue;
		if (ne == SGD_HULL_MAX_EDGES) { rc = -2; goto finish; }
		h->edge_a[ne] = (unsigned char)lo; h->edge_b[ne] = (unsigned char)hi; ++ne;
	}
	h->ne = ne;
	h->aabb_min = sgh_v3(bmin[0], bmin[1], bmin[2]); h->aabb_max = sgh_v3(bmax[0], bmax[1], bmax[2]);
	h->bound_radius = br; h->volume = (float)vol;
	h->unit_inertia = sgh_v3((float)I[0][0], (float)I[1][1], (float)I[2][2]);
	com_out[0] = (float)cm.x; com_out[1] = (float)cm.y; com_out[2] = (float)cm.z;
	// rotation matrix (columns = principal axes) -> quaternion
	{
		const double m00 = V[0][0], m11 = V[1][1], m22 = V[2][2];
		double qw, qx, qy, qz;
		const double tr = m00 + m11 + m22;
		if (tr > 0.0) { const double s = sqrt(tr + 1.0) * 2.0; qw = 0.25 * s; qx = (V[2][1] - V[1][2]) / s; qy = (V[0][2] - V[2][0]) / s; qz = (V[1][0] - V[0][1]) / s; }
		else if (m00 > m11 && m00 > m22) { const double s = sqrt(1.0 + m00 - m11 - m22) * 2.0; qw = (V[2][1] - V[1][2]) / s; qx = 0.25 * s; qy = (V[0][1] + V[1][0]) / s; qz = (V[0][2] + V[2][0]) / s; }
		else if (m11 > m22) { const double s = sqrt(1.0 + m11 - m00 - m22) * 2.0; qw = (V[0][2] - V[2][0]) / s; qx = (V[0][1] + V[1][0]) / s; qy = 0.25 * s; qz = (V[1][2] + V[2][1]) / s; }
		else { const double s = sqrt(1.0 + m22 - m00 - m11) * 2.0; qw = (V[1][0] - V[0][1]) / s; qx = (V[0][2] + V[2][0]) / s; qy = (V[1][2] + V[2][1]) / s; qz = 0.25 * s; }
		rot_out[0] = (float)qx; rot_out[1] = (float)qy; rot_out[2] = (float)qz; rot_out[3] = (float)qw;
	}
	// the two faces of every edge (f0 holds it as a -> b with a < b ... or not: whichever face runs it from edge_a to edge_b)
	for (int e = 0; e < ne; ++e) { h->edge_f0[e] = 0xFFFF; h->edge_f1[e] = 0xFFFF; }
	for (int f = 0; f < nf; ++f) for (int k = fstart[f]; k < fstart[f + 1]; ++k) {
		const int a = fidx[k], b = fidx[k + 1 < fstart[f + 1] ? k + 1 : fstart[f]];
		for (int e = 0; e < ne; ++e) {
			if (h->edge_a[e] == a && h->edge_b[e] == b) { h->edge_f0[e] = (unsigned short)f; break; }
			if (h->edge_a[e] == b && h->edge_b[e] == a) { h->edge_f1[e] = (unsigned short)f; break; }
		}
	}
	// (an edge that only one face runs along -- its neighbour took a point within eps of its plane for a member and lost it as a corner -- stays marked 0xFFFF:
	//  the Gauss-map test does not apply to it, the search evaluates its pairs in full)
	for (int e = 0; e < ne; ++e) if (h->edge_f0[e] == 0xFFFF || h->edge_f1[e] == 0xFFFF) { h->edge_f0[e] = 0xFFFF; h->edge_f1[e] = 0xFFFF; }
	// (test hook, tests/test_big_hull_parity_gpu.py: every 50th edge of a large hull declared open, so that the searches' path for such edges is exercised -- the
	//  builder itself has not produced one since covered faces are dropped)
	/* checks 2 and 3 of 3: a closed surface (every edge between two faces) of the topology of a sphere */
	{
		int open_edges = 0;
		for (int e = 0; e < ne; ++e) if (h->edge_f0[e] == 0xFFFF) ++open_edges;
		if (open_edges || nv - ne + nf != 2) { rc = -3; goto finish; }
	}
	if (ne > 90 && getenv("SGP_HULL_TEST_OPEN_EDGES")) for (int e = 7; e < ne; e += 50) { h->edge_f0[e] = 0xFFFF; h->edge_f1[e] = 0xFFFF; }
	}
finish:
	free(fstart_); free(fidx_); free(fn2); free(fd2);
	}
cleanup:
	free(fn); free(fd); free(fmem_start); free(fmem);
#undef SGH_FAIL
	return rc;
}

/* Returns 0 on success.  com_out / rot_out (quaternion xyzw): body frame expressed in the input frame, i.e.
   input point = com + rot * body point.  com_offset (may be NULL): JPH::OffsetCenterOfMassShape -- the body's centre of mass is
   moved by this vector (input frame) away from the hull's own; the inertia is taken about the moved point. */
static inline int sgd_hull_build(const float* pts_in, int n_in, const float* com_offset, sgd_hull* h, float com_out[3], float rot_out[4])
{
	memset(h, 0, sizeof(*h));
	if (n_in < 4) return -1;
	// 1. unique points
	sgh_d3 pts[256]; int n = 0;
	double ext = 0.0;
	for (int i = 0; i < n_in; ++i) for (int k = 0; k < 3; ++k) { if (!isfinite(pts_in[3 * i + k])) return -1; ext = fmax(ext, fabs((double)pts_in[3 * i + k])); }
	if (!(ext > 0.0)) return -1;
	const double eps = 1.0e-5 * ext;
	if (n_in > 256) {
		/* a larger cloud (a dynamic mesh with thousands of vertices): the extreme points of ALL input points along a fixed set of directions
		   (Fibonacci sphere) plus the six axis directions -- every vertex takes part, wherever it sits in the array; at most 256 of them
		   (JPH::ConvexHullShape::cMaxPointsInHull) */
		int cand[2 * SGD_HULL_MAX_VERTS + 6]; int nc = 0;
		for (int kk = 0; kk < 2 * SGD_HULL_MAX_VERTS + 6 && nc < SGD_HULL_MAX_VERTS; ++kk) {
			sgh_d3 dir;
			if (kk < 6) { dir.x = dir.y = dir.z = 0.0; const double sg = (kk & 1) ? -1.0 : 1.0; if (kk / 2 == 0) dir.x = sg; else if (kk / 2 == 1) dir.y = sg; else dir.z = sg; }
			else {
				const int k = ((kk - 6) * 37) % (2 * SGD_HULL_MAX_VERTS);
				const double z = 1.0 - (2.0 * k + 1.0) / (2.0 * SGD_HULL_MAX_VERTS), rr = sqrt(1.0 - z * z), ph = k * 2.399963229728653;
				dir.x = rr * cos(ph); dir.y = rr * sin(ph); dir.z = z;
			}
			int bi = 0; double bd = -1.0e300;
			for (int i = 0; i < n_in; ++i) { const double d = dir.x * pts_in[3 * i] + dir.y * pts_in[3 * i + 1] + dir.z * pts_in[3 * i + 2]; if (d > bd) { bd = d; bi = i; } }
			int seen = 0;
			for (int j = 0; j < nc; ++j) {
				const sgh_d3 a = { pts_in[3 * bi], pts_in[3 * bi + 1], pts_in[3 * bi + 2] }, b = { pts_in[3 * cand[j]], pts_in[3 * cand[j] + 1], pts_in[3 * cand[j] + 2] };
				const sgh_d3 d = sgh_d3_sub(a, b);
				if (cand[j] == bi || sgh_d3_dot(d, d) < eps * eps) { seen = 1; break; }
			}
			if (!seen) cand[nc++] = bi;
		}
		for (int j = 0; j < nc; ++j) { const sgh_d3 p = { pts_in[3 * cand[j]], pts_in[3 * cand[j] + 1], pts_in[3 * cand[j] + 2] }; pts[n++] = p; }
	} else
	for (int i = 0; i < n_in && n < 256; ++i) {
		const sgh_d3 p = { pts_in[3 * i], pts_in[3 * i + 1], pts_in[3 * i + 2] };
		int dup = 0;
		for (int j = 0; j < n; ++j) { const sgh_d3 d = sgh_d3_sub(p, pts[j]); if (sgh_d3_dot(d, d) < eps * eps) { dup = 1; break; } }
		if (!dup) pts[n++] = p;
	}
	if (n < 4) return -1;
	/* 2 .. 7 on the unique points.  A cloud whose result fails the builder's checks (points coplanar only to rounding, or scattered about a plane by about the
	   tolerance, can defeat either path) is tried again: <= 32 points with every triple as a candidate plane, largest first; then with the tolerance four times
	   as wide, up to 2.6e-3 of the extent (JPH::ConvexHullShapeSettings::mHullTolerance is 1e-3 m: points that close to a face belong to it); a larger cloud that
	   fails at every tolerance is reduced to its 32 extreme points, as every larger cloud was in rounds 1-4, and goes through the same ladder. */
	int rc = -1;
	for (int pass = 0; pass < 2; ++pass) {
		double e = eps;
		for (int k = 0; k < 5; ++k, e *= 4.0) {
			rc = sgd_hull_build_pts(pts, n, e, ext, 0, com_offset, h, com_out, rot_out);
			if (rc == 0) return 0;
			if (n <= SGD_HULL_SMALL_VERTS) { rc = sgd_hull_build_pts(pts, n, e, ext, 1, com_offset, h, com_out, rot_out); if (rc == 0) return 0; }
		}
		if (pass == 1 || n <= SGD_HULL_SMALL_VERTS) break;
		sgh_d3 red[SGD_HULL_SMALL_VERTS]; int nr = 0;
		for (int kk = 0; kk < 2 * SGD_HULL_SMALL_VERTS + 6 && nr < SGD_HULL_SMALL_VERTS; ++kk) {
			sgh_d3 dir;
			if (kk < 6) { dir.x = dir.y = dir.z = 0.0; const double sg = (kk & 1) ? -1.0 : 1.0; if (kk / 2 == 0) dir.x = sg; else if (kk / 2 == 1) dir.y = sg; else dir.z = sg; }
			else {
				const int k = ((kk - 6) * 37) % (2 * SGD_HULL_SMALL_VERTS);
				const double z = 1.0 - (2.0 * k + 1.0) / (2.0 * SGD_HULL_SMALL_VERTS), rr = sqrt(1.0 - z * z), ph = k * 2.399963229728653;
				dir.x = rr * cos(ph); dir.y = rr * sin(ph); dir.z = z;
			}
			int bi = 0; double bd = -1.0e300;
			for (int i = 0; i < n; ++i) { const double dd = sgh_d3_dot(dir, pts[i]); if (dd > bd) { bd = dd; bi = i; } }
			int seen = 0;
			for (int j = 0; j < nr; ++j) if (red[j].x == pts[bi].x && red[j].y == pts[bi].y && red[j].z == pts[bi].z) { seen = 1; break; }
			if (!seen) red[nr++] = pts[bi];
		}
		if (nr < 4) break;
		for (int i = 0; i < nr; ++i) pts[i] = red[i];
		n = nr;
	}
	return rc;
}

// The +-1 cube every box is a scaled copy of (hull id 0 of every world).
static inline void sgd_hull_cube_template(sgd_hull* h)
{
	float pts[24]; int k = 0;
	for (int x = -1; x <= 1; x += 2) for (int y = -1; y <= 1; y += 2) for (int z = -1; z <= 1; z += 2) { pts[k++] = (float)x; pts[k++] = (float)y; pts[k++] = (float)z; }
	float com[3], rot[4];
	sgd_hull_build(pts, 8, NULL, h, com, rot);
	// exact axis-aligned data regardless of what the eigen solver returned for the degenerate (isotropic) inertia
	k = 0;
	for (int i = 0; i < h->nv; ++i) { h->verts[i] = sgh_v3(h->verts[i].x < 0 ? -1.0f : 1.0f, h->verts[i].y < 0 ? -1.0f : 1.0f, h->verts[i].z < 0 ? -1.0f : 1.0f); }
	for (int f = 0; f < h->nf; ++f) {
		v3 n = h->normals[f];
		n = sgh_v3(fabsf(n.x) > 0.5f ? (n.x < 0 ? -1.0f : 1.0f) : 0.0f, fabsf(n.y) > 0.5f ? (n.y < 0 ? -1.0f : 1.0f) : 0.0f, fabsf(n.z) > 0.5f ? (n.z < 0 ? -1.0f : 1.0f) : 0.0f);
		h->normals[f] = n; h->plane_d[f] = 1.0f;
	}
	h->is_box_template = 1;
}

