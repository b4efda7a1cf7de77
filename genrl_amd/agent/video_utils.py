"""MI355X-native `agent/video_utils.py`: the connector (VideoSSM) and aligner (UNetDenoiser) with
the reference's API (mazpie/genrl agent/video_utils.py) on top of the HIP kernels."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import dreamer_utils as common
from .. import noise, ops


class ResidualLinear(nn.Module):  # ref :8-25
    def __init__(self, in_channels, out_channels, norm='layer', act='SiLU', prenorm=False):
        super().__init__()
        assert not prenorm
        self.norm_layer = common.NormLayer(norm, out_channels)
        self.act = common.get_act(act)
        self._plain = (norm == 'none')
        assert self._plain == isinstance(self.act, nn.Identity)
        self.layer = nn.Linear(in_channels, out_channels)
        self.res_proj = nn.Identity() if in_channels == out_channels else nn.Linear(in_channels, out_channels)

    def forward(self, x):
        h = ops.linear(x, self.layer.weight, self.layer.bias)
        if not self._plain:
            ln = self.norm_layer._layer
            h = ops.ln_act(h, ln.weight, ln.bias, ln.eps, act=True)
        r = x if isinstance(self.res_proj, nn.Identity) else ops.linear(x, self.res_proj.weight, self.res_proj.bias)
        return h + r


class UNetDenoiser(nn.Module):  # ref :27-61
    def __init__(self, in_channels, mid_channels, n_layers, norm='layer', act='SiLU'):
        super().__init__()
        out_channels = in_channels
        self.down = nn.ModuleList([ResidualLinear(in_channels, mid_channels if i == n_layers - 1 else in_channels,
                                                  norm=norm, act=act) for i in range(n_layers)])
        self.mid = nn.ModuleList([ResidualLinear(mid_channels, mid_channels, norm=norm, act=act) for _ in range(n_layers)])
        self.up = nn.ModuleList([ResidualLinear(mid_channels * 2, out_channels, norm='none', act='Identity') if i == 0
                                 else ResidualLinear(out_channels * 2, out_channels, norm=norm, act=act)
                                 for i in range(n_layers)])

    def forward(self, x):
        down_res = []
        for layer in self.down:
            x = layer(x)
            down_res.append(x)
        for layer in self.mid:
            x = layer(x)
        down_res.reverse()
        for layer, res in zip(self.up, down_res):
            x = layer(torch.cat([x, res], dim=-1))
        return x


class VideoSSM(common.EnsembleRSSM):  # ref :64-240
    def __init__(self, *args, connector_kl={}, temporal_embeds=False, detached_post=True, n_frames=8,
                 token_dropout=0., loss_scale=1, clip_add_noise=0, clip_lafite_noise=0, rescale_embeds=False,
                 denoising_ae=False, learn_initial=True, **kwargs):
        super().__init__(*args, **kwargs)
        assert not temporal_embeds and token_dropout == 0 and detached_post and learn_initial, \
            'shipped connector configuration only (agent/genrl.yaml:15)'
        self.n_frames = n_frames
        self.viclip_emb_dim = kwargs['action_dim'] - self.n_frames
        self.temporal_embeds, self.detached_post, self.connector_kl = temporal_embeds, detached_post, connector_kl
        self.token_dropout, self.loss_scale, self.rescale_embeds = token_dropout, loss_scale, rescale_embeds
        self.clip_add_noise, self.clip_lafite_noise = clip_add_noise, clip_lafite_noise
        self.clip_const = np.sqrt(self.viclip_emb_dim).item()
        self.denoising_ae = denoising_ae
        if self.denoising_ae:
            self.aligner = UNetDenoiser(self.viclip_emb_dim, self.viclip_emb_dim // 2, n_layers=2, norm='layer', act='SiLU')
        self.learn_initial = learn_initial
        self.initial_state_pred = nn.Sequential(
            nn.Linear(kwargs['action_dim'], kwargs['hidden']),
            common.NormLayer(kwargs['norm'], kwargs['hidden']), common.get_act('SiLU'),
            nn.Linear(kwargs['hidden'], kwargs['hidden']),
            common.NormLayer(kwargs['norm'], kwargs['hidden']), common.get_act('SiLU'),
            nn.Linear(kwargs['hidden'], kwargs['deter']))
        del self._obs_out
        del self._obs_dist

    def _initial_deter(self, init_embed):
        p = self.initial_state_pred
        x = common._dense_ln_silu(init_embed, p[0], p[1])
        x = common._dense_ln_silu(x, p[3], p[4])
        return ops.linear(x, p[6].weight, p[6].bias)

    def initial(self, batch_size, init_embed=None, ignore_learned=False, site='conn.init_q'):  # ref :100-112
        init = super().initial(batch_size)
        if self.learn_initial and not ignore_learned:
            assert init_embed is not None
            if init_embed.shape[-1] == self.viclip_emb_dim:
                init_embed = torch.cat([init_embed, torch.zeros((*init_embed.shape[:-1], 8), device=init_embed.device)], -1)
            init['deter'] = self._initial_deter(init_embed)
            stoch, stats = self.get_stoch_stats_from_deter_state(init, site=site)
            init['stoch'] = stoch
            init.update(stats)
        return init

    def get_action(self, video_embed):  # ref :114-125
        if self.rescale_embeds:
            video_embed = video_embed * self.clip_const
        z = torch.zeros(list(video_embed.shape[:-1]) + [self.n_frames], device=video_embed.device)
        return torch.cat([video_embed, z], dim=-1)

    def update(self, video_embed, wm_post):  # ref :127-207
        nf = self.n_frames
        B, T = video_embed.shape[:2]
        S, K = self._stoch, self._discrete
        dev = self.device
        metrics = {}
        loss = 0
        ve = video_embed[:, nf - 1::nf].to(dev)
        ve = ve.reshape(B, T // nf, 1, -1).repeat(1, 1, nf, 1).reshape(B, T, -1)
        orig = ve
        if self.clip_add_noise > 0:
            ve = F.normalize(ve + noise.draw('normal', 'conn.add_eps', ve.shape, dev) * self.clip_add_noise, dim=-1)
        if self.clip_lafite_noise > 0:
            nn_ = F.normalize(noise.draw('normal', 'conn.clip_eps', ve.shape, dev), dim=-1)
            ve = F.normalize((1 - self.clip_lafite_noise) * ve + self.clip_lafite_noise * nn_, dim=-1)
        if self.denoising_ae:
            assert (self.clip_lafite_noise + self.clip_add_noise) > 0, 'Nothing to denoise'
            den = F.normalize(self.aligner(ve), dim=-1)
            denoising_loss = 1 - F.cosine_similarity(den, orig, dim=-1).mean()
            loss = loss + denoising_loss
            metrics['aligner_cosine_distance'] = denoising_loss
            ve = orig
        acts = self.get_action(ve).transpose(0, 1).contiguous()                    # (T,B,520)
        post = {k: v.reshape(B, T, *v.shape[2:]).detach() for k, v in wm_post.items()}
        # teacher-forced prior rollout: stoch_{t-1} from the world model, deter from the connector.
        # Non-recurrent work batched over T; the recurrence is ops.gru_seq.
        init = self.initial(B, init_embed=acts[0])
        post_stoch_tm = post['stoch'].reshape(B, T, S * K).transpose(0, 1)
        prev = torch.cat([init['stoch'].reshape(1, B, S * K), post_stoch_tm[:-1]], 0).reshape(T * B, S * K)
        x = common._dense_ln_silu(prev, self._img_in[0], self._img_in[1], acts.reshape(T * B, -1))
        deter = ops.gru_seq(x.reshape(T, B, -1), None, init['deter'], self._cell._layer.weight,
                            self._cell._norm.weight, self._cell._norm.bias)
        prior_logit = self._prior_logits(deter.reshape(T * B, -1)).reshape(T, B, S, K).transpose(0, 1)
        kl_loss, kl_value = self.kl_loss(post, {'logit': prior_logit}, **self.connector_kl)
        loss = loss + self.loss_scale * kl_loss
        metrics['connector_kl'] = kl_value.mean()
        # initial KL (metric only, ref :197-205)
        with torch.no_grad():
            G = T // nf
            ve2 = ve.reshape(B, G, nf, -1)[:, 1:, 0].reshape(B * (G - 1), -1)
            a2 = self.get_action(ve2)
            post2 = {k: v.reshape(B, G, nf, *v.shape[2:])[:, 1:, 0].reshape(B * (G - 1), *v.shape[2:]) for k, v in post.items()}
            prev_state = self.initial(B * (G - 1), init_embed=a2, site='conn.ikl_init_q')
            prior2 = self.img_step(prev_state, a2, site='conn.ikl_step_q')
            _, ikl = self.kl_loss(post2, prior2, **self.connector_kl)
            metrics['connector_initial_kl'] = ikl.mean()
        return loss, metrics

    def video_imagine(self, video_embed, dreamer_init=None, sample=True, reset_every_n_frames=True, denoise=False):
        """ref :209-240"""
        nf = self.n_frames
        B, T = video_embed.shape[:2]
        if self.denoising_ae and denoise:
            video_embed = F.normalize(self.aligner(video_embed), dim=-1)
        action = self.get_action(video_embed)
        init = self.initial(batch_size=B, init_embed=action[:, 0], site='imag.target_init_q')
        if dreamer_init is not None:
            init[self.cell_input] = dreamer_init[self.cell_input]
        if reset_every_n_frames:
            chunks = []
            for action_chunk in torch.chunk(action, T // nf, dim=1):
                prior = self.imagine(action_chunk, init, sample=sample)
                chunks.append(prior)
                init = self.initial(batch_size=B, ignore_learned=True)
                init[self.cell_input] = prior[self.cell_input][:, -1]
            return {k: torch.cat([c[k] for c in chunks], dim=1) for k in chunks[0]}
        return self.imagine(action, init, sample=sample)
