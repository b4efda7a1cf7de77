"""MI355X-native `agent/genrl.py`: GenRLAgent and the connector update hook with the reference's
API (mazpie/genrl agent/genrl.py)."""
import torch

from .dreamer import DreamerAgent, ActorCritic, stop_gradient, env_reward
from . import dreamer_utils as common
from . import video_utils
from .. import streams
from ..tools.genrl_utils import *          # reward fns looked up through globals(), ref :5,122


def connector_update_fn(self, module_name, data, outputs, metrics):  # ref :7-25
    connector = getattr(self, module_name)
    if not getattr(self.cfg, 'viclip_encode', False):
        raise NotImplementedError('on-the-fly InternVideo2 embedding is host-side preprocessing; provide '
                                  "data['clip_video'] (viclip_encode=True), as process_dataset.py does")
    return connector.update(data['clip_video'], outputs['post'])


class GenRLAgent(DreamerAgent):  # ref :27-124
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.n_frames = 8
        self.viclip_emb_dim = 512
        assert self.cfg.batch_length % self.n_frames == 0, 'Fix batch length param'
        if 'clip_video' in self.obs_space:
            self.viclip_emb_dim = self.obs_space['clip_video'].shape[0]
        connector = video_utils.VideoSSM(**self.cfg.connector, **self.cfg.connector_rssm,
                                         connector_kl=self.cfg.connector_kl, n_frames=self.n_frames,
                                         action_dim=self.viclip_emb_dim + self.n_frames,
                                         clip_add_noise=self.cfg.clip_add_noise,
                                         clip_lafite_noise=self.cfg.clip_lafite_noise,
                                         device=self.device, cell_input='stoch')
        connector.to(self.device)
        connector.requires_grad_(False)
        self.wm.add_module_to_update('connector', connector, connector_update_fn,
                                     detached=self.cfg.connector.detached_post)
        if getattr(self.cfg, 'imag_reward_fn', None) is not None:
            self.instantiate_imag_behavior()

    def instantiate_imag_behavior(self):
        self._imag_behavior = ActorCritic(self.cfg, self.act_spec, self.wm.inp_size, name='imag').to(self.device)
        self._imag_behavior.requires_grad_(False)
        self._imag_behavior.rewnorm = common.StreamNorm(**self.cfg.imag_reward_norm, device=self.device)

    def finetune_mode(self):
        self._acting_behavior = self._imag_behavior
        self.wm.detached_update_fns = {}
        self.wm.e2e_update_fns = {}
        self.wm.grad_heads.append('reward')

    def update_wm(self, data, step):
        return super().update_wm(data, step)

    def report(self, data, key='observation', nvid=8):  # ref :64-106
        with torch.no_grad():
            n_frames = self.wm.connector.n_frames
            obs = data['observation'][:nvid, n_frames:]
            B, T = obs.shape[:2]
            report_data = super().report(data)
            wm = self.wm
            truth = data[key][:nvid].float() / 255
            decoder = wm.heads['decoder']
            pre = wm.preprocess(data)
            embed = wm.encoder(pre)
            states, _ = wm.rssm.observe(embed[:nvid, :n_frames], data['action'][:nvid, :n_frames],
                                        data['is_first'][:nvid, :n_frames])
            recon = decoder(wm.decoder_input_fn(states))[key].mean[:nvid]
            dreamer_init = {k: v[:, -1] for k, v in states.items()}
            assert getattr(self.cfg, 'viclip_encode', False)
            video_embed = data['clip_video'][:nvid, n_frames * 2 - 1::n_frames].to(self.device)
            video_embed = video_embed.reshape(B, T // n_frames, -1).unsqueeze(2).repeat(1, 1, n_frames, 1).reshape(B, T, -1)
            prior = wm.connector.video_imagine(video_embed, dreamer_init, reset_every_n_frames=False)
            prior_recon = decoder(wm.decoder_input_fn(prior))[key].mean
            model = torch.clip(torch.cat([recon[:, :n_frames] + 0.5, prior_recon + 0.5], 1), 0, 1)
            error = (model - truth + 1) / 2
            report_data['video_clip_pred'] = torch.cat([truth, model, error], 3)
        return report_data

    def update_imag_behavior(self, state=None, outputs=None, metrics={}, seq_data=None):  # ref :108-124
        if getattr(self.cfg, 'imag_reward_fn', None) is None:
            return outputs['post'], metrics
        if outputs is not None:
            post, is_terminal = outputs['post'], outputs['is_terminal']
        else:
            seq_data = self.wm.preprocess(seq_data)
            with torch.no_grad():
                post, _ = self.wm.rssm.observe(self.wm.encoder(seq_data), seq_data['action'], seq_data['is_first'])
            is_terminal = seq_data['is_terminal']
        start = {k: stop_gradient(v) for k, v in post.items()}
        imag_reward_fn = lambda seq: globals()[self.cfg.imag_reward_fn](self, seq, **self.cfg.imag_reward_args)
        metrics.update(self._imag_behavior.update(self.wm, start, is_terminal, imag_reward_fn))
        streams.join()          # side-stream connector updates (cfg.overlap_detached) are ordered from here on
        return start, metrics
