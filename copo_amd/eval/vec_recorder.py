"""Vectorised counterpart of the reference's `RecorderEnv` (copo/eval/recoder.py:73-349): the per-episode evaluation row of
`eval/evaluate_population.py` -- one row per WHOLE scene episode, the same column names and the same definitions -- computed
from the sampler's dense `[T, E, N]` tensors for E scenes at once.

Definitions restated from the reference (what is averaged over what):
  velocity_step_mean_episode_{min,mean,max}   per env step: mean velocity over the agents that acted (`"step_reward" in info`,
                                              recoder.py:124-126); then min / mean / max over the steps that had one (:198-209)
  num_neighbours_mean_episode_{mean,max}      per env step: mean neighbour count (within `neighbours_distance`, the recorder's
                                              default is 20 m, :76) over every agent that received a reward -- acting agents AND
                                              the ones spawned in that step (:107-114); then mean / max over steps (:225-237)
  num_agents_total, *_per_300_steps           agents that terminated in the episode; env episode length = number of steps (:244-252)
  success_rate / crash_rate / out_rate        over those agents (:139-152, :249, :279-285)
  episode_reward_{mean,min,max}               last `info["episode_reward"]` of every agent (:254-262)
  episode_cost_{mean,min,max,sum}             per-agent sum of `info["cost"]` (:264-274)
  episode_length_mean, success_episode_length_mean   last `info["episode_length"]` of every agent / of the successful ones (:287-299)
Agents still driving when the episode ends (forced end after 5 x horizon steps) count like the reference counts them: their
last row carries the done flag.  The energy columns and the SVO estimate of the recorder are left out (MetaDrive's energy
model is out of scope).

Device-side running sums per scene, one host read per finished episode batch."""
import numpy as np
import torch

F_ACTED, F_DONE, F_ARRIVE, F_CRASH, F_OUT, F_MAXSTEP, F_SPAWNED, F_ENV_RESET = (1 << i for i in range(8))
I_VELOCITY, I_COST, I_EPISODE_LENGTH, I_EPISODE_REWARD = 0, 4, 5, 6

COLUMNS = ("velocity_step_mean_episode_min", "velocity_step_mean_episode_mean", "velocity_step_mean_episode_max",
           "num_neighbours_mean_episode_mean", "num_neighbours_mean_episode_max", "num_agents_total",
           "num_agents_total_per_300_steps", "success_rate", "num_agents_success", "num_agents_success_per_300_steps",
           "num_agents_failed_per_300_steps", "episode_reward_mean", "episode_reward_min", "episode_reward_max",
           "episode_cost_mean", "episode_cost_min", "episode_cost_max", "episode_cost_sum", "crash_rate", "num_agents_crash",
           "out_rate", "num_agents_out", "episode_length_mean", "success_episode_length_mean", "env_episode_steps")


class VecRecorder:
    def __init__(self, E, device):
        self.E, self.device = int(E), device
        f64 = torch.float64
        z = lambda: torch.zeros(self.E, dtype=f64, device=device)  # noqa: E731
        inf = lambda s: torch.full((self.E,), s * float("inf"), dtype=f64, device=device)  # noqa: E731
        self.acc = dict(steps=z(), vsteps=z(), vsum=z(), vmin=inf(1), vmax=inf(-1), nsteps=z(), nsum=z(), nmax=inf(-1),
                        agents=z(), succ=z(), crash=z(), out=z(), rsum=z(), rmin=inf(1), rmax=inf(-1), csum=z(), cmin=inf(1),
                        cmax=inf(-1), lsum=z(), slsum=z())
        self._init = {k: v.clone() for k, v in self.acc.items()}
        self.rows = []
        self.cost_acc = None      # [E, N] per-agent sum of info["cost"] over its steps (recoder.py:264-274), sized on first use

    def _flush(self, mask):
        """Finish the episode of the scenes in `mask` [E] bool: emit their rows, re-arm their accumulators."""
        idx = mask.nonzero(as_tuple=False).view(-1)
        if idx.numel() == 0:
            return
        a = {k: v[idx].cpu().numpy() for k, v in self.acc.items()}
        for i in range(idx.numel()):
            n, steps = a["agents"][i], max(a["steps"][i], 1.0)
            if n <= 0:
                continue
            self.rows.append({
                "velocity_step_mean_episode_min": a["vmin"][i], "velocity_step_mean_episode_mean": a["vsum"][i] / max(a["vsteps"][i], 1.0),
                "velocity_step_mean_episode_max": a["vmax"][i],
                "num_neighbours_mean_episode_mean": a["nsum"][i] / max(a["nsteps"][i], 1.0), "num_neighbours_mean_episode_max": a["nmax"][i],
                "num_agents_total": n, "num_agents_total_per_300_steps": n / steps * 300.0, "success_rate": a["succ"][i] / n,
                "num_agents_success": a["succ"][i], "num_agents_success_per_300_steps": a["succ"][i] / steps * 300.0,
                "num_agents_failed_per_300_steps": a["crash"][i] / steps * 300.0,
                "episode_reward_mean": a["rsum"][i] / n, "episode_reward_min": a["rmin"][i], "episode_reward_max": a["rmax"][i],
                "episode_cost_mean": a["csum"][i] / n, "episode_cost_min": a["cmin"][i], "episode_cost_max": a["cmax"][i],
                "episode_cost_sum": a["csum"][i], "crash_rate": a["crash"][i] / n, "num_agents_crash": a["crash"][i],
                "out_rate": a["out"][i] / n, "num_agents_out": a["out"][i], "episode_length_mean": a["lsum"][i] / n,
                "success_episode_length_mean": a["slsum"][i] / a["succ"][i] if a["succ"][i] > 0 else 0.0,
                "env_episode_steps": a["steps"][i], "scene": int(idx[i])})
        for k, v in self.acc.items():
            v[idx] = self._init[k][idx]

    def add(self, batch, keep=None):
        """Consume one sampler fragment.  `keep` [T, E] bool: steps that count (scenes outside their evaluated episodes are
        masked out by the caller); None = all."""
        fl = batch["flags"].to(torch.int32)                     # [T, E, N]
        info, nbr = batch["infos"], batch["nbr_cnt"]
        T = fl.shape[0]
        A = self.acc
        f64 = torch.float64
        for t in range(T):
            f = fl[t]
            k = torch.ones(self.E, dtype=torch.bool, device=self.device) if keep is None else keep[t]
            acted = (f & F_ACTED) > 0
            # (the row that carries F_ENV_RESET flags the NEXT episode's slots as spawned; their neighbour counts belong to the
            #  scene before the reset and the reference's RecorderEnv never sees a reset observation: not part of this episode)
            resetting = ((f & F_ENV_RESET) > 0).any(-1, keepdim=True)
            present = acted | (((f & F_SPAWNED) > 0) & ~resetting)
            na, npres = acted.sum(-1).to(f64), present.sum(-1).to(f64)
            vel = (info[t, :, :, I_VELOCITY].to(f64) * acted).sum(-1) / na.clamp(min=1.0)
            has_v = k & (na > 0)
            A["steps"] += k.to(f64)
            A["vsteps"] += has_v.to(f64)
            A["vsum"] += torch.where(has_v, vel, torch.zeros_like(vel))
            A["vmin"] = torch.where(has_v, torch.minimum(A["vmin"], vel), A["vmin"])
            A["vmax"] = torch.where(has_v, torch.maximum(A["vmax"], vel), A["vmax"])
            nn_ = (nbr[t].to(f64) * present).sum(-1) / npres.clamp(min=1.0)
            has_n = k & (npres > 0)
            A["nsteps"] += has_n.to(f64)
            A["nsum"] += torch.where(has_n, nn_, torch.zeros_like(nn_))
            A["nmax"] = torch.where(has_n, torch.maximum(A["nmax"], nn_), A["nmax"])
            if self.cost_acc is None:
                self.cost_acc = torch.zeros(f.shape, dtype=f64, device=self.device)
            self.cost_acc += info[t, :, :, I_COST].to(f64) * (acted & k[:, None])
            done = acted & ((f & F_DONE) > 0) & k[:, None]
            all_done = acted & ((f & F_DONE) > 0)
            if bool(done.any()):
                d64 = done.to(f64)
                succ = done & ((f & F_ARRIVE) > 0)
                er, el = info[t, :, :, I_EPISODE_REWARD].to(f64), info[t, :, :, I_EPISODE_LENGTH].to(f64)
                ec = self.cost_acc          # the agent's cost summed over its steps
                big = torch.full_like(er, float("inf"))
                A["agents"] += d64.sum(-1)
                A["succ"] += succ.to(f64).sum(-1)
                A["crash"] += (done & ((f & F_CRASH) > 0)).to(f64).sum(-1)
                A["out"] += (done & ((f & F_OUT) > 0)).to(f64).sum(-1)
                A["rsum"] += (er * d64).sum(-1)
                A["rmin"] = torch.minimum(A["rmin"], torch.where(done, er, big).amin(-1))
                A["rmax"] = torch.maximum(A["rmax"], torch.where(done, er, -big).amax(-1))
                A["csum"] += (ec * d64).sum(-1)
                A["cmin"] = torch.minimum(A["cmin"], torch.where(done, ec, big).amin(-1))
                A["cmax"] = torch.maximum(A["cmax"], torch.where(done, ec, -big).amax(-1))
                A["lsum"] += (el * d64).sum(-1)
                A["slsum"] += (el * succ.to(f64)).sum(-1)
            self.cost_acc = torch.where(all_done, torch.zeros_like(self.cost_acc), self.cost_acc)
            ended = ((f & F_ENV_RESET) > 0).any(-1) & k
            if bool(ended.any()):
                self._flush(ended)

    def frame(self):
        import pandas as pd
        return pd.DataFrame(self.rows)

    def means(self):
        if not self.rows:
            return {}
        return {c: float(np.mean([r[c] for r in self.rows])) for c in COLUMNS}
