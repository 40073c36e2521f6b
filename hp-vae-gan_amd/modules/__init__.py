"""Mirror of the reference's `modules` package: networks_3d, networks_2d, losses, utils."""
