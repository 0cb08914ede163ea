/*
 * geo_oracle.c -- CPU restatement of the reference's WGS84 <-> ECEF <-> ENU transforms
 * (/root/reference/uavPathPlanning.cpp:893-1108, constants uavPathPlanning.hpp:133-173).
 * TEST INFRASTRUCTURE ONLY (see dense_oracle.c).  PINNED: the reference's README prints the ENU
 * images and the WGS84 round trip of seven waypoints to 15 decimals (readme.md:10-28); those are
 * committed as tests/golden/G1_readme_geo.json and this file reproduces them.
 *
 * Point layouts: lla = {lon_deg, lat_deg, alt_m} (struct WGS84Point), enu = {east, north, up}.
 */
#define _GNU_SOURCE   /* M_PI under -std=c11 */
#include <math.h>

#define WGS84_A 6378137.0
#define WGS84_E2 0.006694379990141

static double deg2rad(double d) { return d * M_PI / 180.0; }
static double rad2deg(double r) { return r * 180.0 / M_PI; }
static double prime_vertical_radius(double lat_rad) {          /* calcN, hpp:139-142 */
    double s = sin(lat_rad);
    return WGS84_A / sqrt(1.0 - WGS84_E2 * s * s);
}

static void lla_to_ecef(const double *lla, double *ecef) {     /* cpp:894-910 */
    double lat = deg2rad(lla[1]), lon = deg2rad(lla[0]);
    double N = prime_vertical_radius(lat);
    ecef[0] = (N + lla[2]) * cos(lat) * cos(lon);
    ecef[1] = (N + lla[2]) * cos(lat) * sin(lon);
    ecef[2] = (N * (1 - WGS84_E2) + lla[2]) * sin(lat);
}

static void ecef_to_lla(const double *e, double *lla) {        /* cpp:926-968, fixed-point iteration */
    double p = sqrt(e[0] * e[0] + e[1] * e[1]);
    double theta = atan2(e[2] * WGS84_A, p * WGS84_A * (1 - WGS84_E2));
    double lat = atan2(e[2] + WGS84_E2 * WGS84_A * (1 - WGS84_E2) * pow(sin(theta), 3) / (1 - WGS84_E2),
                       p - WGS84_E2 * WGS84_A * pow(cos(theta), 3));
    for (int i = 0; i < 10; ++i) {
        double N = prime_vertical_radius(lat);
        double alt = p / cos(lat) - N;
        double nl = atan2(e[2], p * (1 - WGS84_E2 * N / (N + alt)));
        int done = fabs(nl - lat) < 1e-12;
        lat = nl;
        if (done) break;
    }
    double N = prime_vertical_radius(lat);
    lla[0] = rad2deg(atan2(e[1], e[0]));
    lla[1] = rad2deg(lat);
    lla[2] = (p < 1e-12) ? fabs(e[2]) - WGS84_A * sqrt(1 - WGS84_E2) : p / cos(lat) - N;
}

/* wgs84ToENU (cpp:1046-1063) for n targets against one reference point */
int csp_oracle_wgs84_to_enu(const double *lla, const double *ref, double *enu, long n) {
    double r0[3];
    lla_to_ecef(ref, r0);
    double lat = deg2rad(ref[1]), lon = deg2rad(ref[0]);
    double cl = cos(lat), sl = sin(lat), co = cos(lon), so = sin(lon);
    for (long i = 0; i < n; ++i) {
        double t[3];
        lla_to_ecef(lla + 3 * i, t);
        double dx = t[0] - r0[0], dy = t[1] - r0[1], dz = t[2] - r0[2];
        enu[3 * i + 0] = -so * dx + co * dy + 0.0 * dz;                  /* rows of cpp:976-1000 */
        enu[3 * i + 1] = -sl * co * dx + -sl * so * dy + cl * dz;
        enu[3 * i + 2] = cl * co * dx + cl * so * dy + sl * dz;
    }
    return 0;
}

/* enuToWGS84 (cpp:1066-1083) */
int csp_oracle_enu_to_wgs84(const double *enu, const double *ref, double *lla, long n) {
    double r0[3];
    lla_to_ecef(ref, r0);
    double lat = deg2rad(ref[1]), lon = deg2rad(ref[0]);
    double cl = cos(lat), sl = sin(lat), co = cos(lon), so = sin(lon);
    for (long i = 0; i < n; ++i) {
        const double *q = enu + 3 * i;
        double t[3];
        t[0] = r0[0] + (-so * q[0] + -sl * co * q[1] + cl * co * q[2]);  /* transpose, cpp:1003-1027 */
        t[1] = r0[1] + (co * q[0] + -sl * so * q[1] + cl * so * q[2]);
        t[2] = r0[2] + (0.0 * q[0] + cl * q[1] + sl * q[2]);
        ecef_to_lla(t, lla + 3 * i);
    }
    return 0;
}
