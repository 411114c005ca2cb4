// placeholder until the SGM oracle lands
