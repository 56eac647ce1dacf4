"""Host-side mirror of the reference's `modules` package for the VAE training path: same module
names, symbols, argument meaning and error behaviour, with the arithmetic delegated to
libsgvae.so (MI355X).  Put `simulgen-vae_amd/` on sys.path (or use
`simulgen_vae_amd.install_reference_api()`) and `from modules.VAE_network import VAE`,
`from modules.train import train`, ... resolve here instead of in the reference."""
