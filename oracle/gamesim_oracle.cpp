// gamesim_oracle.cpp -- CPU restatement of robo-game-sim's physics step.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the product path (robo-game-sim_amd/)
// may include, link, import or call this file; it is the checker for the HIP
// stepper (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
//
// PARITY STATUS: "parity unpinned" against a live reference build.  The
// reference (src/GameSim.cpp) needs Eigen3, which this image does not have,
// and it ships no tests / golden vectors, so it cannot be
// run here.  The only reference outputs available are the sample values the
// survey session recorded in SURVEY.md section 8c; tests/test_oracle_kat.py pins
// this file to those (17 significant digits).  Everything else is a careful
// restatement, statement by statement, of the cited reference lines.
//
// What is restated (all citations relative to /root/reference):
//   reset                    src/GameSim.cpp:21-55
//   undecided / run          src/GameSim.cpp:57-60, 118-124
//   step (updateSim)         src/GameSim.cpp:126-171
//   fill_grid                src/GameSim.cpp:173-217   (populateStateGrid)
//   accel_* / rk4            src/GameSim.cpp:62-116    (f_player/f_puck/RK4_*)
//   resolve_first            src/GameSim.cpp:219-358   (carryOutFirstCollision)
//   scan_for_events          src/GameSim.cpp:360-511   (insertCollisions)
//   disc_radius / disc_mass  src/GameSim.cpp:513-542   (getEntityInfo)
//   push_apart / push_inside src/GameSim.cpp:581-623   (correctOverlap x2)
//   Event + ordering         include/Collision.h:5-30
//   state vector layout      include/SimState.h:14-23
//   log record               src/GameSim.cpp:161-170 + include/utils.h:64-76
//
// Generalisation (no reference counterpart; the reference is hard-wired 2v2):
// n players per team, discs ordered A1..An, B1..Bn, PK; state vector
// [TAS, TBS, PK(4), A1(4)..An(4), B1(4)..Bn(4)]; spawn rows evenly spaced in y.
// At n = 2 every formula reduces to the reference's.
//
// Floating point: IEEE binary64, one rounding per operation in the reference's
// association order.  Build with -ffp-contract=off and without -ffast-math.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

namespace {

constexpr int kGrid = 50;        // COLLISION_GRID_POINTS   include/GameSim.h:6
constexpr int kLoopCap = 50;     // COLLISION_COUNTER_LIM   include/GameSim.h:7
constexpr double kPushBuffer = 1.1;  // OVERLAP_BUFFER      include/GameSim.h:9
constexpr int kMaxTeam = 7;
constexpr int kMaxDiscs = 2 * kMaxTeam + 1;

enum Kind { UP = 0, DOWN = 1, RIGHT = 2, LEFT = 3, PAIR = 4 };  // Collision.h:5

struct Event {                   // include/Collision.h:7-17
  double t = 1.0e9;
  int i = -1;
  int j = -1;
  int kind = -1;
};

struct Later {                   // include/Collision.h:24-30: earliest at the BACK
  bool operator()(const Event &a, const Event &b) const { return a.t > b.t; }
};

struct Vec4 { double px, py, vx, vy; };

struct Stats {
  long long steps = 0;           // executed (undecided) steps
  long long resolved = 0;        // events carried out
  long long cap_hits = 0;        // steps that ended on the 50-iteration cap
  long long base_overflow = 0;   // base index > 50 (reference: out of bounds)
  long long skipped_after_goal = 0;  // cpp:230-231
  long long steps_with_events = 0;
  long long hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // resolved per step: 0,1,2,3,4-7,8-15,16-31,32+
};

struct Arena {
  int n = 2, players = 4, discs = 5, pk = 4;
  // arena constants, src/GameSim.cpp:5-9
  double X = 10.0, Y = 5.0, R_player = 0.2, R_puck = 0.175, goal_h = 1.0;
  // parameters, src/GameSim.cpp:27-46
  double dt = 0.05, dt_col = 0.001, t = 0.0, sigma = 0.0;
  int win = 3;
  double tau_puck = 0.1, tau_player = 0.5, m_player = 1.0, m_puck = 0.5;
  // state
  double TAS = 0.0, TBS = 0.0;
  Vec4 x[kMaxDiscs];
  double damage[2 * kMaxTeam][2 * kMaxTeam];
  bool scored = false;
  // per-step grid (state_grid_, include/GameSim.h:46): S, T and U rows
  Vec4 S[kGrid + 1][kMaxDiscs];
  double T[kGrid + 1];
  double U[kMaxDiscs][2];
  // noise, src/GameSim.cpp:3,41,66-67
  std::default_random_engine reng;
  std::normal_distribution<double> w_dist{0.0, 1.0};
  bool draw_always = false;      // true: draw 2 normals per accel_player like the reference
  Stats st;
  std::vector<Event> list;       // live events of the current step
  // optional log file, cpp:48-53,161-170
  FILE *log = nullptr;
};

inline double disc_radius(const Arena &a, int e) { return e == a.pk ? a.R_puck : a.R_player; }
inline double disc_mass(const Arena &a, int e) { return e == a.pk ? a.m_puck : a.m_player; }

// f_player, src/GameSim.cpp:62-68.  With sigma == 0 the two draws only add
// +-0.0, so they may be skipped without changing any value (zero sign aside).
inline void accel_player(Arena &a, Vec4 &d, const Vec4 &x, const double u[2]) {
  d.px = x.vx;
  d.py = x.vy;
  if (a.draw_always || a.sigma != 0.0) {
    d.vx = (u[0] - x.vx) / a.tau_player + a.sigma * a.w_dist(a.reng);
    d.vy = (u[1] - x.vy) / a.tau_player + a.sigma * a.w_dist(a.reng);
  } else {
    d.vx = (u[0] - x.vx) / a.tau_player;
    d.vy = (u[1] - x.vy) / a.tau_player;
  }
}

// f_puck, src/GameSim.cpp:90-96
inline void accel_puck(const Arena &a, Vec4 &d, const Vec4 &x, const double u[2]) {
  d.px = x.vx;
  d.py = x.vy;
  d.vx = (u[0] - 0.2 * x.vx) / a.tau_puck;
  d.vy = (u[1] - 0.2 * x.vy) / a.tau_puck;
}

inline Vec4 axpy(const Vec4 &x, const Vec4 &k, double h) {  // x + k*h, cpp:75-76
  return {x.px + k.px * h, x.py + k.py * h, x.vx + k.vx * h, x.vy + k.vy * h};
}

// RK4_player / RK4_puck + simAgnostic dispatch, src/GameSim.cpp:70-116,549-555
Vec4 rk4(Arena &a, int e, const Vec4 &x, const double u[2], double h) {
  Vec4 k1, k2, k3, k4;
  const bool puck = (e == a.pk);
  const double half = h / 2.0;
  if (puck) accel_puck(a, k1, x, u); else accel_player(a, k1, x, u);
  Vec4 x2 = axpy(x, k1, half);
  if (puck) accel_puck(a, k2, x2, u); else accel_player(a, k2, x2, u);
  Vec4 x3 = axpy(x, k2, half);
  if (puck) accel_puck(a, k3, x3, u); else accel_player(a, k3, x3, u);
  Vec4 x4 = axpy(x, k3, h);
  if (puck) accel_puck(a, k4, x4, u); else accel_player(a, k4, x4, u);
  const double w = h / 6.0;      // cpp:87,115: x + (k1 + k2*2 + k3*2 + k4) * (dt/6)
  Vec4 r;
  r.px = x.px + (((k1.px + k2.px * 2.0) + k3.px * 2.0) + k4.px) * w;
  r.py = x.py + (((k1.py + k2.py * 2.0) + k3.py * 2.0) + k4.py) * w;
  r.vx = x.vx + (((k1.vx + k2.vx * 2.0) + k3.vx * 2.0) + k4.vx) * w;
  r.vy = x.vy + (((k1.vy + k2.vy * 2.0) + k3.vy * 2.0) + k4.vy) * w;
  return r;
}

// populateStateGrid, src/GameSim.cpp:173-217
void fill_grid(Arena &a, const double *controls) {
  for (int e = 0; e < a.discs; ++e) {
    if (e == a.pk) { a.U[e][0] = 0.0; a.U[e][1] = 0.0; }
    else { a.U[e][0] = controls[2 * e]; a.U[e][1] = controls[2 * e + 1]; }
    a.S[0][e] = a.x[e];
  }
  a.T[0] = a.t;
  for (int i = 1; i <= kGrid; ++i) {
    a.T[i] = a.t + a.dt_col * i;                 // cpp:201
    for (int e = 0; e < a.discs; ++e)            // discs in entity order, cpp:203-215
      a.S[i][e] = rk4(a, e, a.S[i - 1][e], a.U[e], a.dt_col);
  }
}

// correctOverlap (pair), src/GameSim.cpp:581-592.  pij is NOT normalised.
void push_apart(Arena &a, int i, int j, int g, double ri, double rj) {
  Vec4 &si = a.S[g][i];
  Vec4 &sj = a.S[g][j];
  const double pijx = sj.px - si.px, pijy = sj.py - si.py;
  const double overlap = ri + rj - std::sqrt(pijx * pijx + pijy * pijy);
  if (overlap > 0) {
    const double ci = kPushBuffer * ri / (ri + rj) * overlap;
    const double cj = kPushBuffer * rj / (ri + rj) * overlap;
    si.px -= ci * pijx; si.py -= ci * pijy;
    sj.px += cj * pijx; sj.py += cj * pijy;
  }
}

// correctOverlap (wall), src/GameSim.cpp:594-623
void push_inside(Arena &a, int i, int g, double ri, int kind) {
  Vec4 &s = a.S[g][i];
  double overlap = 0.0, nx = 0.0, ny = 0.0;
  switch (kind) {
    case UP:    overlap =  s.py + ri - a.Y / 2.0; nx = 0.;  ny = -1.; break;
    case DOWN:  overlap = -s.py + ri - a.Y / 2.0; nx = 0.;  ny = 1.;  break;
    case RIGHT: overlap =  s.px + ri - a.X / 2.0; nx = -1.; ny = 0.;  break;
    case LEFT:  overlap = -s.px + ri - a.X / 2.0; nx = 1.;  ny = 0.;  break;
  }
  if (overlap > 0) {
    const double c = kPushBuffer * overlap;
    s.px += c * nx;
    s.py += c * ny;
  }
}

// carryOutFirstCollision, src/GameSim.cpp:219-358.  Returns the discs to re-scan.
int resolve_first(Arena &a, std::vector<Event> &list, int &base, int checks[2]) {
  const double damage_coeff = 0.01;              // cpp:221
  Event c = list.back();                         // cpp:227
  list.pop_back();
  if ((c.i == a.pk || c.j == a.pk) && a.scored) {  // cpp:230-231
    a.st.skipped_after_goal++;
    return 0;
  }
  base = static_cast<int>(std::floor((c.t - a.t) / a.dt_col)) + 1;  // cpp:237
  if (base > kGrid) {            // reference would index past the grid here (UB); clamp + count
    a.st.base_overflow++;
    base = kGrid;
  }
  if (base < 1) base = 1;
  a.st.resolved++;

  if (c.kind == PAIR) {                          // cpp:242-299
    const int i = c.i, j = c.j;
    const double mi = disc_mass(a, i), mj = disc_mass(a, j);
    const double ri = disc_radius(a, i), rj = disc_radius(a, j);
    const double dt_t = c.t - a.T[base - 1];     // cpp:254
    Vec4 xi = rk4(a, i, a.S[base - 1][i], a.U[i], dt_t);
    Vec4 xj = rk4(a, j, a.S[base - 1][j], a.U[j], dt_t);
    {                                            // cpp:259-265
      const double pijx = xj.px - xi.px, pijy = xj.py - xi.py;
      const double overlap = ri + rj - std::sqrt(pijx * pijx + pijy * pijy);
      if (overlap > 0) {
        const double ci = 1.01 * ri / (ri + rj) * overlap;
        const double cj = 1.01 * rj / (ri + rj) * overlap;
        xi.px -= ci * pijx; xi.py -= ci * pijy;
        xj.px += cj * pijx; xj.py += cj * pijy;
      }
    }
    const double p1x = xi.px, p1y = xi.py, v1x = xi.vx, v1y = xi.vy;  // cpp:268-271
    const double p2x = xj.px, p2y = xj.py, v2x = xj.vx, v2y = xj.vy;
    {                                            // cpp:273
      const double dx = p1x - p2x, dy = p1y - p2y;
      const double s = 2 * mj / (mi + mj) * ((v1x - v2x) * dx + (v1y - v2y) * dy) / (dx * dx + dy * dy);
      xi.vx = v1x - s * dx;
      xi.vy = v1y - s * dy;
    }
    {                                            // cpp:274
      const double dx = p2x - p1x, dy = p2y - p1y;
      const double s = 2 * mi / (mi + mj) * ((v2x - v1x) * dx + (v2y - v1y) * dy) / (dx * dx + dy * dy);
      xj.vx = v2x - s * dx;
      xj.vy = v2y - s * dy;
    }
    if (i != a.pk && j != a.pk) {                // cpp:277-282
      const double dvx = v1x - v2x, dvy = v1y - v2y;
      const double dv_sq = damage_coeff * (dvx * dvx + dvy * dvy);
      a.damage[i][j] += mi * dv_sq;
      a.damage[j][i] += mj * dv_sq;
    }
    a.S[base][i] = rk4(a, i, xi, a.U[i], a.dt_col - dt_t);   // cpp:287-289
    a.S[base][j] = rk4(a, j, xj, a.U[j], a.dt_col - dt_t);
    push_apart(a, i, j, base, ri, rj);
    for (int g = base + 1; g <= kGrid; ++g) {    // cpp:292-297
      a.S[g][i] = rk4(a, i, a.S[g - 1][i], a.U[i], a.dt_col);
      a.S[g][j] = rk4(a, j, a.S[g - 1][j], a.U[j], a.dt_col);
      push_apart(a, i, j, g, ri, rj);
    }
  } else {                                       // wall, cpp:300-343
    const int i = c.i;
    const double ri = disc_radius(a, i);
    const double dt_t = c.t - a.T[base - 1];     // cpp:308
    Vec4 xi = rk4(a, i, a.S[base - 1][i], a.U[i], dt_t);
    switch (c.kind) {                            // cpp:314-317
      case UP:    if (xi.vy > 0) xi.vy *= -1.0; break;
      case DOWN:  if (xi.vy < 0) xi.vy *= -1.0; break;
      case RIGHT: if (xi.vx > 0) xi.vx *= -1.0; break;
      case LEFT:  if (xi.vx < 0) xi.vx *= -1.0; break;
    }
    if (i == a.pk && (c.kind == LEFT || c.kind == RIGHT) &&
        xi.py < a.goal_h / 2.0 && xi.py > -a.goal_h / 2.0) {   // cpp:321-327
      if (!a.scored) { if (c.kind == RIGHT) a.TAS++; else a.TBS++; }
      a.S[base][i] = {0.0, 0.0, 0.0, 0.0};
      a.scored = true;
    } else {                                     // cpp:330-332
      a.S[base][i] = rk4(a, i, xi, a.U[i], a.dt_col - dt_t);
      push_inside(a, i, base, ri, c.kind);
    }
    for (int g = base + 1; g <= kGrid; ++g) {    // cpp:336-340
      a.S[g][i] = rk4(a, i, a.S[g - 1][i], a.U[i], a.dt_col);
      push_inside(a, i, g, ri, c.kind);
    }
  }

  // cpp:347-352: drop every event that touches c.i or c.j
  list.erase(std::remove_if(list.begin(), list.end(),
                            [&c](const Event &e) {
                              return (c.i == e.i) || (c.i == e.j && e.j != -1) ||
                                     (c.j == e.i) || (c.j == e.j && e.j != -1);
                            }),
             list.end());
  int n = 0;
  checks[n++] = c.i;                             // cpp:355-356
  if (c.j != -1) checks[n++] = c.j;
  return n;
}

// insertCollisions, src/GameSim.cpp:360-511
void scan_for_events(Arena &a, const int *checks, int n_checks, std::vector<Event> &list, int base) {
  for (int c = 0; c < n_checks; ++c) {
    const int id = checks[c];
    const double rad = disc_radius(a, id);
    for (int g = base; g < kGrid + 1; ++g) {     // base >= 1, so the reference's i==0 arms are dead
      Event best;
      const Vec4 &k = a.S[g][id];
      const Vec4 &m = a.S[g - 1][id];
      const double t_m = a.T[g - 1];
      if (k.py + rad >= a.Y / 2.0) {             // cpp:374-394
        if (m.vy > 0) {
          const double tau = std::fmax(0.0, (a.Y / 2.0 - rad - m.py) / m.vy);
          const double tc = t_m + tau;
          if (tc < best.t) { best.t = tc; best.i = id; best.j = -1; best.kind = UP; }
        }
      }
      if (k.py - rad <= -a.Y / 2.0) {            // cpp:397-417
        if (m.vy < 0) {
          const double tau = std::fmax(0.0, (-a.Y / 2.0 + rad - m.py) / m.vy);
          const double tc = t_m + tau;
          if (tc < best.t) { best.t = tc; best.i = id; best.j = -1; best.kind = DOWN; }
        }
      }
      if (k.px + rad >= a.X / 2.0) {             // cpp:420-440
        if (m.vx > 0) {
          const double tau = std::fmax(0.0, (a.X / 2.0 - rad - m.px) / m.vx);
          const double tc = t_m + tau;
          if (tc < best.t) { best.t = tc; best.i = id; best.j = -1; best.kind = RIGHT; }
        }
      }
      if (k.px - rad <= -a.X / 2.0) {            // cpp:443-463
        if (m.vx < 0) {
          const double tau = std::fmax(0.0, (-a.X / 2.0 + rad - m.px) / m.vx);
          const double tc = t_m + tau;
          if (tc < best.t) { best.t = tc; best.i = id; best.j = -1; best.kind = LEFT; }
        }
      }
      for (int o = 0; o < a.discs; ++o) {        // cpp:466-498, entity order
        if (o == id) continue;
        const Vec4 &ok = a.S[g][o];
        const double rad_o = disc_radius(a, o);
        const double ddx = k.px - ok.px, ddy = k.py - ok.py;
        if (std::sqrt(ddx * ddx + ddy * ddy) <= rad + rad_o) {
          const Vec4 &om = a.S[g - 1][o];
          const double dpx = om.px - m.px, dpy = om.py - m.py;     // cpp:484
          const double dvx = om.vx - m.vx, dvy = om.vy - m.vy;     // cpp:485
          const double sigma = rad + rad_o;
          const double dpdv = dpx * dvx + dpy * dvy;
          const double dvdv = dvx * dvx + dvy * dvy;
          const double dpdp = dpx * dpx + dpy * dpy;
          const double d = dpdv * dpdv - dvdv * (dpdp - sigma * sigma);  // cpp:487
          const double dvdp = dvx * dpx + dvy * dpy;
          if (dvdp < 0 && d >= 0) {
            const double tau = std::fmax(0.0, -(dvdp + std::sqrt(d)) / dvdv);  // cpp:490
            const double tc = t_m + tau;
            if (tc < best.t) { best.t = tc; best.i = id; best.j = o; best.kind = PAIR; }
          }
        }
      }
      if (best.t - a.t < a.dt) {                 // cpp:501-505
        list.push_back(best);
        break;
      }
    }
  }
  // cpp:510.  Tie order matters (equal-time events are common: tau clamps to 0) and std::sort gives no guarantee in
  // general; the reference's list never holds more than E <= 15 events, and libstdc++'s std::sort is a plain
  // insertion sort below 16 elements, hence stable: of equal-time events the later-inserted ends up nearer the
  // back and is popped first (SURVEY Appendix B checked this against std::stable_sort on the compiled reference).
  std::sort(list.begin(), list.end(), Later());
}

// updateSim, src/GameSim.cpp:126-171
void step(Arena &a, const double *controls) {
  a.scored = false;
  int base = 1;
  fill_grid(a, controls);
  std::vector<Event> &list = a.list;
  list.clear();
  int all[kMaxDiscs];
  for (int e = 0; e < a.discs; ++e) all[e] = e;
  int counter = 0;
  const long long resolved_before = a.st.resolved;
  do {                                           // cpp:140-150
    int two[2];
    if (!list.empty()) {
      const int n = resolve_first(a, list, base, two);
      scan_for_events(a, two, n, list, base);
    } else {
      scan_for_events(a, all, a.discs, list, base);
    }
    counter++;
  } while (!list.empty() && counter < kLoopCap);
  if (!list.empty()) a.st.cap_hits++;
  for (int e = 0; e < a.discs; ++e) a.x[e] = a.S[kGrid][e];   // cpp:153-157
  a.t += a.dt;                                                // cpp:158

  a.st.steps++;
  const long long r = a.st.resolved - resolved_before;
  if (r > 0) a.st.steps_with_events++;
  const int bin = r < 4 ? (int)r : r < 8 ? 4 : r < 16 ? 5 : r < 32 ? 6 : 7;
  a.st.hist[bin]++;

  if (a.log) {                                   // cpp:161-170: 29 doubles at n = 2
    // t, TAS, TBS, puck xy, each player's xy, then "damage.row(k)" written as
    // players*players/.. contiguous doubles from &damage(k,0) of a COLUMN-major
    // matrix (utils.h:75): row k's record is flat elements k .. k+players-1.
    const int P = a.players;
    std::vector<double> rec;
    rec.push_back(a.t); rec.push_back(a.TAS); rec.push_back(a.TBS);
    rec.push_back(a.x[a.pk].px); rec.push_back(a.x[a.pk].py);
    for (int e = 0; e < P; ++e) { rec.push_back(a.x[e].px); rec.push_back(a.x[e].py); }
    std::vector<double> colmajor(P * P + P);     // +P: the last rows read past the matrix
    for (int r2 = 0; r2 < P; ++r2)
      for (int c2 = 0; c2 < P; ++c2) colmajor[c2 * P + r2] = a.damage[r2][c2];
    for (int k = 0; k < P; ++k)
      for (int q = 0; q < P; ++q) rec.push_back(colmajor[k + q]);
    fwrite(rec.data(), sizeof(double), rec.size(), a.log);
  }
}

void write_state(const Arena &a, double *out) {  // SimState.h:14-23 layout
  out[0] = a.TAS;
  out[1] = a.TBS;
  const Vec4 &p = a.x[a.pk];
  out[2] = p.px; out[3] = p.py; out[4] = p.vx; out[5] = p.vy;
  for (int e = 0; e < a.players; ++e) {
    out[6 + 4 * e + 0] = a.x[e].px; out[6 + 4 * e + 1] = a.x[e].py;
    out[6 + 4 * e + 2] = a.x[e].vx; out[6 + 4 * e + 3] = a.x[e].vy;
  }
}

void do_reset(Arena &a, double dt, int win, const double x0_ball[4], double noise, int seed,
              double tau_puck, double tau_player, double m_player, double m_puck) {
  a.dt = dt;                                     // cpp:27-33
  a.dt_col = dt / kGrid;
  a.t = 0.0;
  a.sigma = noise;
  a.win = win;
  a.TAS = 0;
  a.TBS = 0;
  a.x[a.pk] = {x0_ball[0], x0_ball[1], x0_ball[2], x0_ball[3]};   // cpp:34
  // cpp:35-38 at n = 2: A at x=-X/4, B at x=+X/4, y = +-Y/4.  General n: rows at
  // y_k = Y/2 - (2k+1)*Y/(2n) (exactly +-Y/4 for n = 2).
  for (int k = 0; k < a.n; ++k) {
    const double y = a.Y / 2.0 - (2 * k + 1) * a.Y / (2.0 * a.n);
    a.x[k] = {-a.X / 4.0, y, 0.0, 0.0};
    a.x[a.n + k] = {a.X / 4.0, y, 0.0, 0.0};
  }
  std::memset(a.damage, 0, sizeof(a.damage));    // cpp:39
  a.reng = std::default_random_engine(seed);     // cpp:41 (w_dist deliberately NOT reset)
  a.tau_puck = tau_puck;                         // cpp:43-46
  a.tau_player = tau_player;
  a.m_player = m_player;
  a.m_puck = m_puck;
}

inline bool is_undecided(const Arena &a) { return (a.TAS < a.win) && (a.TBS < a.win); }  // cpp:57-60

}  // namespace

extern "C" {

void *rgso_create(int n_per_team, int draw_always) {
  if (n_per_team < 1 || n_per_team > kMaxTeam) return nullptr;
  Arena *a = new Arena();
  a->n = n_per_team;
  a->players = 2 * n_per_team;
  a->discs = a->players + 1;
  a->pk = a->players;
  a->draw_always = draw_always != 0;
  const double zero[4] = {0, 0, 0, 0};
  do_reset(*a, 0.05, 3, zero, 0.0, 0, 0.1, 0.5, 1.0, 0.5);
  return a;
}

void rgso_destroy(void *h) {
  Arena *a = static_cast<Arena *>(h);
  if (a->log) fclose(a->log);
  delete a;
}

int rgso_state_size(void *h) { return 2 + 4 * static_cast<Arena *>(h)->discs; }

// The reference's noise source as the host's <random> produces it (include/GameSim.h:72-73: std::default_random_engine
// + std::normal_distribution<double>; libstdc++): the first n_normals draws of a generator seeded with `seed`, and the
// first n_canonicals values of generate_canonical<double, 53> from a second engine with the same seed.  Checker for the
// HIP stepper's device generator (csrc/gsim_rng.hpp).
void rgso_rng_streams(long long seed, long long n_normals, double *normals, long long n_canonicals, double *canonicals) {
  std::default_random_engine e1(static_cast<std::default_random_engine::result_type>(seed));
  std::normal_distribution<double> d(0.0, 1.0);
  for (long long i = 0; i < n_normals; ++i) normals[i] = d(e1);
  std::default_random_engine e2(static_cast<std::default_random_engine::result_type>(seed));
  for (long long i = 0; i < n_canonicals; ++i) canonicals[i] = std::generate_canonical<double, 53>(e2);
}

void rgso_reset(void *h, double dt, int winning_score, const double *x0_ball, double noise,
                int log, const char *logname, int seed, double tau_puck, double tau_player,
                double player_mass, double puck_mass, double *out_state) {
  Arena &a = *static_cast<Arena *>(h);
  do_reset(a, dt, winning_score, x0_ball, noise, seed, tau_puck, tau_player, player_mass, puck_mass);
  if (a.log) { fclose(a.log); a.log = nullptr; }         // cpp:48-53
  if (log && logname) a.log = fopen(logname, "wb");      // silent on failure like ofstream
  if (out_state) write_state(a, out_state);
}

int rgso_undecided(void *h) { return is_undecided(*static_cast<Arena *>(h)) ? 1 : 0; }

// run, src/GameSim.cpp:118-124.  controls: players x 2 doubles, disc order.
void rgso_run(void *h, const double *controls, double *out_state) {
  Arena &a = *static_cast<Arena *>(h);
  if (is_undecided(a)) step(a, controls);
  if (a.log) fflush(a.log);
  if (out_state) write_state(a, out_state);
}

void rgso_get_damage(void *h, double *out) {
  Arena &a = *static_cast<Arena *>(h);
  for (int i = 0; i < a.players; ++i)
    for (int j = 0; j < a.players; ++j) out[i * a.players + j] = a.damage[i][j];
}

double rgso_get_time(void *h) { return static_cast<Arena *>(h)->t; }

// Extension for tests: overwrite the whole state (reference can only set the puck).
void rgso_set_state(void *h, const double *state, double t) {
  Arena &a = *static_cast<Arena *>(h);
  a.TAS = state[0];
  a.TBS = state[1];
  a.x[a.pk] = {state[2], state[3], state[4], state[5]};
  for (int e = 0; e < a.players; ++e)
    a.x[e] = {state[6 + 4 * e], state[7 + 4 * e], state[8 + 4 * e], state[9 + 4 * e]};
  a.t = t;
}

// out[0..13]: steps, resolved, cap_hits, base_overflow, skipped_after_goal, steps_with_events, hist[8]
void rgso_get_stats(void *h, long long *out) {
  const Stats &s = static_cast<Arena *>(h)->st;
  out[0] = s.steps; out[1] = s.resolved; out[2] = s.cap_hits; out[3] = s.base_overflow;
  out[4] = s.skipped_after_goal; out[5] = s.steps_with_events;
  for (int k = 0; k < 8; ++k) out[6 + k] = s.hist[k];
}

// Batched driver used as the parity checker for the HIP stepper and as the CPU
// baseline.  Layouts match the device side (arena index fastest):
//   x0_ball   [4][n_arenas]
//   controls  [n_steps][players][2][n_arenas]   (ctrl_step_stride = 0 => same controls every step)
//   state_out [S][n_arenas]                      final state, SimState.h order along S
//   traj_out  [n_rec][S][n_arenas]               state after steps rec_every, 2*rec_every, ... (may be null)
//   damage_out[players*players][n_arenas]        (may be null)
//   stats_out [14]                               summed over arenas (may be null)
// Arenas [a_begin, a_end) are processed by n_threads OpenMP threads.
int rgso_batch_run(int n_per_team, long long n_arenas, long long a_begin, long long a_end, int n_steps,
                   double dt, int winning_score, double noise, int seed_base, double tau_puck,
                   double tau_player, double player_mass, double puck_mass, const double *x0_ball,
                   const double *controls, long long ctrl_step_stride, double *state_out,
                   int rec_every, double *traj_out, double *damage_out, long long *stats_out,
                   int n_threads, int draw_always) {
  if (n_per_team < 1 || n_per_team > kMaxTeam) return -1;
  const int P = 2 * n_per_team, E = P + 1, S = 2 + 4 * E;
  long long tot[14] = {0};
  if (n_threads < 1) n_threads = 1;
#pragma omp parallel num_threads(n_threads)
  {
    Arena *ap = new Arena();
    Arena &a = *ap;
    a.n = n_per_team; a.players = P; a.discs = E; a.pk = P;
    a.draw_always = draw_always != 0;
    std::vector<double> u(2 * P), st(S);
    long long loc[14] = {0};
#pragma omp for schedule(dynamic, 16)
    for (long long ar = a_begin; ar < a_end; ++ar) {
      const double x0[4] = {x0_ball[0 * n_arenas + ar], x0_ball[1 * n_arenas + ar],
                            x0_ball[2 * n_arenas + ar], x0_ball[3 * n_arenas + ar]};
      a.w_dist.reset();
      a.st = Stats();
      do_reset(a, dt, winning_score, x0, noise, seed_base + (int)ar, tau_puck, tau_player,
               player_mass, puck_mass);
      for (int k = 0; k < n_steps; ++k) {
        const double *ck = controls + (long long)k * ctrl_step_stride;
        for (int p = 0; p < P; ++p) {
          u[2 * p + 0] = ck[((long long)p * 2 + 0) * n_arenas + ar];
          u[2 * p + 1] = ck[((long long)p * 2 + 1) * n_arenas + ar];
        }
        if (is_undecided(a)) step(a, u.data());
        if (traj_out && rec_every > 0 && (k + 1) % rec_every == 0) {
          write_state(a, st.data());
          const long long r = (k + 1) / rec_every - 1;
          for (int s = 0; s < S; ++s) traj_out[(r * S + s) * n_arenas + ar] = st[s];
        }
      }
      write_state(a, st.data());
      for (int s = 0; s < S; ++s) state_out[(long long)s * n_arenas + ar] = st[s];
      if (damage_out)
        for (int i = 0; i < P; ++i)
          for (int j = 0; j < P; ++j) damage_out[((long long)i * P + j) * n_arenas + ar] = a.damage[i][j];
      loc[0] += a.st.steps; loc[1] += a.st.resolved; loc[2] += a.st.cap_hits;
      loc[3] += a.st.base_overflow; loc[4] += a.st.skipped_after_goal; loc[5] += a.st.steps_with_events;
      for (int q = 0; q < 8; ++q) loc[6 + q] += a.st.hist[q];
    }
#pragma omp critical
    for (int q = 0; q < 14; ++q) tot[q] += loc[q];
    delete ap;
  }
  if (stats_out) for (int q = 0; q < 14; ++q) stats_out[q] = tot[q];
  return 0;
}

}  // extern "C"
