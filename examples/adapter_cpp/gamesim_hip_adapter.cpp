// The adapter of INTEGRATION.md "Option B" as a stand-alone program: a C++ class with the public surface of the
// reference's GameSim (include/GameSim.h:16-27: reset / undecided / run; std::array where the reference has Eigen vectors,
// so that it builds without Eigen) on top of the C ABI of include/rgs_hip.h, and a main() that plays one game with it.
//
//   g++ -O2 -I include examples/adapter_cpp/gamesim_hip_adapter.cpp -o adapter -L robo-game-sim_amd/csrc -lrgs_hip \
//       -Wl,-rpath,$PWD/robo-game-sim_amd/csrc
//   ./adapter <logfile> <noise> <seed> <steps>      prints the final state, one value per line (%.17g), then undecided()
//
// tests/test_gpu_adapter.py builds and runs it on the GPU box and compares state and log file with the oracle's.
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "rgs_hip.h"

class GameSim {                                     // same public surface as include/GameSim.h:16-27
 public:
  using State = std::array<double, 22>;
  using Vec2 = std::array<double, 2>;
  GameSim() { check(rgs_create(1, 2, 0, &h_)); check(rgs_device_alloc(h_, 8, &ctl_dev_)); }
  ~GameSim() { rgs_destroy(h_); }                   // (closes an open log and frees the handle's device buffers)
  State reset(double dt, int winning_score, const std::array<double, 4> &x0_ball, double noise, bool log,
              const std::string &logname, int seed, double tau_puck, double tau_player, double player_mass, double puck_mass) {
    check(rgs_reset(h_, dt, winning_score, x0_ball.data(), /*x0_batched=*/0, noise, seed, tau_puck, tau_player, player_mass,
                    puck_mass));                    // noise != 0: the reference's generator, on the device (bit-identical)
    log_ = log;                                     // src/GameSim.cpp:48-53: open the log, or close the one that is open
    const int arena = 0;
    const char *path = logname.c_str();
    check(log ? rgs_log_open(h_, 1, &arena, &path) : rgs_log_close(h_));
    return state();
  }
  bool undecided() { uint8_t u; check(rgs_undecided(h_, &u)); return u != 0; }
  State run(const Vec2 &a1, const Vec2 &a2, const Vec2 &b1, const Vec2 &b2) {
    const double u[8] = {a1[0], a1[1], a2[0], a2[1], b1[0], b1[1], b2[0], b2[1]};   // [P][2][B=1]
    if (!log_) {
      check(rgs_step(h_, u));
    } else {                                        // cpp:160-170: the step's 29-double record goes to the log file
      check(rgs_device_upload(h_, ctl_dev_, u, 8));
      check(rgs_step_device_log(h_, ctl_dev_, 0, 1));
    }
    return state();
  }

 private:
  State state() { State s; check(rgs_get_state(h_, s.data())); return s; }
  static void check(int rc) { if (rc != RGS_OK) throw std::runtime_error(rgs_last_error()); }
  rgs_sim *h_ = nullptr;
  double *ctl_dev_ = nullptr;
  bool log_ = false;
};

int main(int argc, char **argv) {
  if (argc < 5) { std::fprintf(stderr, "usage: %s <logfile> <noise> <seed> <steps>\n", argv[0]); return 2; }
  const std::string logname = argv[1];
  const double noise = std::atof(argv[2]);
  const int seed = std::atoi(argv[3]), steps = std::atoi(argv[4]);
  try {
    GameSim sim;
    GameSim::State s = sim.reset(0.05, 2, {3.5, 0.2, 6.0, 0.0}, noise, true, logname, seed, 0.1, 0.5, 1.0, 0.5);
    s = sim.run({0, 0}, {0, 0}, {0, 0}, {0, 0});                            // py/MonteCarlo.py:33
    double t = 0.0;
    for (int k = 0; k < steps; ++k) {                                         // py/run_sim.py:62-65: everybody (cos t, sin t); a decided game: run() is a no-op
      const GameSim::Vec2 u = {std::cos(t), std::sin(t)};
      s = sim.run(u, u, u, u);
      t += 0.05;
    }
    const bool open_game = sim.undecided();
    sim.reset(0.05, 2, {0, 0, 0, 0}, 0.0, false, "", 0, 0.1, 0.5, 1.0, 0.5);  // closes the log like the reference's next reset
    for (double v : s) std::printf("%.17g\n", v);
    std::printf("%d\n", open_game ? 1 : 0);
  } catch (const std::exception &e) {
    std::fprintf(stderr, "adapter: %s\n", e.what());
    return 1;
  }
  return 0;
}
