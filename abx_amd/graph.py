"""hipGraph capture of the reverse-diffusion step.

One step = ScoreNetwork call (3 passes, ~600 kernel launches through ctypes) + get_prev + FullDiffuser.reverse.  At the bench's
100 samples per GPU the launches are invisible; at the 12-13 samples per GPU of 8-way strong scaling, or a single sample, the host
cannot issue them as fast as the GPU retires them.  Every entry of the C ABI is capture-safe (no internal synchronisation, no
allocation), so a whole step is recorded once into a graph and replayed.

What makes the step replayable:
  * `t` and the step index live in static device tensors that are updated before each replay (the Philox counter of the reverse
    kernel reads the step through AbxReverseArgs.step_dev, not a launch-time constant);
  * the carried state (rigids_t, seq_t) is copied back into static tensors at the end of the captured step;
  * the self-conditioning tensors ping-pong between the model's two representation buffers and a 3-pass call flips their role, so
    TWO graphs are captured (for even and odd steps) and replayed alternately;
  * all per-step outputs are allocated during capture from the graph's private pool and keep their addresses.
Injected noise (`noise_fn`, parity tests) is not supported: graphs run with the device Philox generator only."""
import torch

from abx_amd import ops, sampler
from abx_amd.model.abx import get_prev


class GraphedSteps:
    def __init__(self, batch, config, diffuser, model, diffuse_mask, dt, sample_ids, center=True, noise_scale=1.0, guidance=None):
        self.batch, self.cfg, self.D, self.model = batch, config, diffuser, model
        self.dm, self.dt, self.sid, self.center, self.noise_scale, self.guidance = diffuse_mask, dt, sample_ids, center, noise_scale, guidance
        dev = batch['rigids_t'].device
        B = batch['rigids_t'].shape[0]
        self.t = torch.zeros(B, device=dev, dtype=torch.float64)
        self.step = torch.zeros(1, device=dev, dtype=torch.int32)
        self.ones = torch.ones(B, device=dev, dtype=torch.float32)
        self.rig = batch['rigids_t'].to(torch.float64).clone()
        self.seq = batch['seq_t'].to(torch.int64).clone()
        self.graphs = {}                 # parity -> (graph, out, post_state)
        self.pool = None

    def _body(self, first=False):
        b = self.batch
        if torch.cuda.is_current_stream_capturing():
            ops.range_words(self.rig.device).zero_()              # (captured: every replay starts from a clear range word)
        if not first:        # step 0 keeps the batch's own float32 rigids_t (sample_ref): the reference's step-0 dtype quirk
            b['rigids_t'], b['seq_t'] = self.rig, self.seq
        sampler.set_t_feats(b, self.D, self.t, self.ones)
        out = self.model(b)
        f = out['heads']['folding']
        if self.cfg.model.heads.diffusion_module.embed.embed_self_conditioning:
            b.update(get_prev(b, out, self.cfg.model))
        rot_score, trans_score = f['rot_score'], f['trans_score']
        if self.guidance is not None:
            rot_score, trans_score = self.guidance(b, out, rot_score, trans_score, self.dm)
        rig, seq = self.D.reverse(rigid_t=b['rigids_t'], seq_t=b['seq_t'], rot_score=rot_score, trans_score=trans_score,
                                  logits_t=out['heads']['sequence_module']['logits'], diffuse_mask=self.dm, t=self.t, dt=self.dt,
                                  center=self.center, noise_scale=self.noise_scale, sample_ids=self.sid, step=0, step_dev=self.step)
        self.rig.copy_(rig)
        self.seq.copy_(seq)
        return out

    def run(self, k, t):
        """Step k at time t: eager the first time a parity is met AFTER an eager warm-up step, replayed afterwards.
        Returns the model output dict of the step (static tensors: clone what must outlive the next step of the same parity)."""
        self.t.fill_(float(t))
        self.step.fill_(int(k))
        parity = k & 1
        b = self.batch
        if k == 0 or torch.cuda.is_current_stream_capturing():
            out = self._body(first=True)                         # eager: lazy initialisations, workspaces, kernel attributes
            b['rigids_t'], b['seq_t'] = self.rig, self.seq
            return out
        if parity not in self.graphs:
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, pool=self.pool):
                out = self._body()
            self.pool = g.pool()
            post = {kk: b[kk] for kk in ('prev_pos', 'prev_seq', 'prev_pair', 't', 'rot_score_scaling', 'trans_score_scaling', 'is_recycling')
                    if kk in b}
            self.graphs[parity] = (g, out, post)
        g, out, post = self.graphs[parity]
        g.replay()
        # range safety of the split-f16 kernels: the eager path repeats a flagged pass on the exact kernels (model/abx.py); a replayed
        # graph cannot branch, and the step has overwritten its own self-conditioning input by now, so here it is an error with the remedy
        bits = 0
        if ops.RANGE_CHECK:
            for v in ops.range_words(self.rig.device).tolist():       # (one word per network pass: ops.RANGE_SLOT)
                bits |= v
        if bits:
            raise FloatingPointError(f'step {k}: an activation left the operand range of the split-f16 kernels ({", ".join(ops.range_names(bits))}); '
                                     'run this trajectory without use_graph: the eager path repeats such a pass on the exact fp32 kernels')
        b.update(post)
        b['rigids_t'], b['seq_t'] = self.rig, self.seq
        return out
