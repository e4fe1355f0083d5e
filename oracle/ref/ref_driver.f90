! TEST INFRASTRUCTURE — not part of the product.
!
! C-callable driver around the reference's OWN per-ray procedures, which are
! compiled unmodified from /root/reference/src by oracle/Makefile:
!   ring, point            (reference src/sourceMod.f90:250, :12)
!   glass_bottle%forward   (src/lens.f90:230)
!   telescope              (src/optics_system.f90:6) -> plano/doublet forward
!   makeImage              (src/imageMod.f90:19)
!   constructors           (src/lens.f90:73,129,170)
! The reference's `program raytrace` (src/main.f90) cannot be linked into a
! library, so this file re-creates the few set-up lines of main.f90:51-81 and
! :113-116 and the two loop bodies main.f90:90-109 / :127-162 as glue around
! those procedures.  Everything numerical on the per-ray path is reference code.
module ortref_api

    use iso_c_binding
    use iso_fortran_env, only : int64
    use vector_class
    use lensMod
    use source,      only : point, ring, create_spot, point_on_bottle, iSORS, emit_image, init_emit_image
    use imageMod,    only : makeImage
    use opticsystem, only : telescope
    use stackMod,    only : stack
    use setup,       only : iris, iris_radius, use_tracker, fibre_offset
    use constants,   only : pi

    implicit none

    type(plano_convex),       save :: L2a, L2b   ! settings wavelength / 843 nm
    type(achromatic_doublet), save :: L3a, L3b
    type(glass_bottle),       save :: bot
    real,    save :: cosThetaMax, r1, r2, img_plane_1, img_diam, besselDiameter, distance
    logical, save :: use_bottle_s
    ! source type: 0 point, 1 spot, 2 crs, 3 isors (setupMod.f90:85-99); 5 = iSORS(ring=.false.) (tests)
    integer, save :: source_s = 0, nphotons_s = 100
    real,    save :: isors_offset_s = 0., ring_width_s = 0., spot_size_s = 0.
    ! image source (source_s == 4): histogram built by init_emit_image, and the working copy
    ! emit_image decrements (main.f90:88 firstprivate(imgin))
    integer, save :: imgin_s(512, 512) = 0, imgwork_s(512, 512) = 0

    interface
        subroutine ortref_rng_table(u, stride, len, first_draw) bind(C, name="ortref_rng_table")
            import :: c_ptr, c_int64_t, c_int32_t
            type(c_ptr), value :: u
            integer(c_int64_t), value :: stride
            integer(c_int32_t), value :: len, first_draw
        end subroutine
        subroutine ortref_rng_key(seed, phase, ray, first_draw) bind(C, name="ortref_rng_key")
            import :: c_int64_t, c_int32_t
            integer(c_int64_t), value :: seed, ray
            integer(c_int32_t), value :: phase, first_draw
        end subroutine
        function ortref_rng_draws() bind(C, name="ortref_rng_draws") result(k)
            import :: c_int32_t
            integer(c_int32_t) :: k
        end function
    end interface

contains

    function cstr(s) result(f)
        character(kind=c_char), intent(IN) :: s(*)
        character(len=:), allocatable :: f
        integer :: n, i
        n = 0
        do while (s(n+1) /= c_null_char)
            n = n + 1
        end do
        allocate(character(len=n) :: f)
        do i = 1, n
            f(i:i) = s(i)
        end do
    end function cstr

    ! Mirrors setupMod.f90:57-133 (values arrive as arguments instead of a
    ! settings file) and main.f90:51-81, :113-116.
    function ortref_init(bottle_path, l2_path, l3_path, wavelength, alpha_deg, n_axicon, &
                         ring_width, image_diameter, fibre_off, iris_mode, iris_rad, use_bottle) &
                         bind(C, name="ortref_init") result(rc)
        character(kind=c_char), intent(IN) :: bottle_path(*), l2_path(*), l3_path(*)
        real(c_double), value :: wavelength, alpha_deg, n_axicon, ring_width, image_diameter
        real(c_double), value :: fibre_off, iris_rad
        integer(c_int), value :: iris_mode, use_bottle
        integer(c_int) :: rc
        real :: alpha, angle, wl2

        alpha = alpha_deg * pi / 180.                     ! setupMod.f90:61
        bot = glass_bottle(cstr(bottle_path), wavelength) ! setupMod.f90:115
        L2a = plano_convex(cstr(l2_path), wavelength)     ! setupMod.f90:117
        L3a = achromatic_doublet(cstr(l3_path), wavelength, 2.*L2a%fb + L2a%thickness) ! :119
        wl2 = 843d-9                                      ! main.f90:114
        L2b = plano_convex(cstr(l2_path), wl2)            ! main.f90:115
        L3b = achromatic_doublet(cstr(l3_path), wl2, 2.*L2b%fb + L2b%thickness) ! main.f90:116

        iris = [iris_mode == 1, iris_mode == 2]           ! setupMod.f90:103-111
        iris_radius = iris_rad
        fibre_offset = fibre_off
        use_tracker = .false.
        use_bottle_s = use_bottle /= 0
        img_diam = image_diameter

        angle = atan(L2a%radius / L2a%fb)                 ! main.f90:51-52
        cosThetaMax = cos(angle)
        if (L2a%fb <= bot%radiusa + bot%centre%z) then    ! main.f90:54-58
            bot%centre%z = L2a%fb - bot%radiusa - 2d-3
        end if
        distance = (bot%radiusa + bot%centre%z)           ! main.f90:63
        besselDiameter = distance*97.3d-3*tan(alpha*(n_axicon - 1)) / (L2a%fb) ! main.f90:66
        r1 = besselDiameter - ring_width                  ! main.f90:68-70
        r2 = (besselDiameter / 2.d0)**2
        r1 = r1**2
        img_plane_1 = 2.*(L2a%fb + L3a%fb) + L2a%thickness + L3a%thickness ! main.f90:81
        rc = 0
    end function ortref_init

    ! Source selection + the source-dependent set-up lines: setupMod.f90:85-99, :132-136 and
    ! main.f90:60-70.  Call after ortref_init.
    subroutine ortref_set_source(source, nphotons, isors_offset, crs_spot_size, alpha_deg, n_axicon, ring_width) &
               bind(C, name="ortref_set_source")
        integer(c_int), value :: source, nphotons
        real(c_double), value :: isors_offset, crs_spot_size, alpha_deg, n_axicon, ring_width
        real :: offset, alpha
        source_s = source
        nphotons_s = nphotons
        isors_offset_s = isors_offset
        ring_width_s = ring_width
        offset = bot%radiusa + bot%centre%z                         ! setupMod.f90:135
        spot_size_s = (crs_spot_size*(L2a%fb - offset)) / L2a%fb    ! setupMod.f90:136
        alpha = alpha_deg * pi / 180.
        if (source == 3 .or. source == 5) then                      ! main.f90:60-64
            distance = bot%radiusa + isors_offset
        else
            distance = (bot%radiusa + bot%centre%z)
        end if
        besselDiameter = distance*97.3d-3*tan(alpha*(n_axicon - 1)) / (L2a%fb)
        r1 = besselDiameter - ring_width
        r2 = (besselDiameter / 2.d0)**2
        r1 = r1**2
    end subroutine ortref_set_source

    ! image source: setupMod.f90:120-121 -> init_emit_image (sourceMod.f90:363-408) with the
    ! 262144 rounding draws keyed as (seed, phase 0, ray 0, draw k); serial semantics
    ! (nphotonsLocal = nphotons: one thread).  counts_out receives imgin(512,512).
    subroutine ortref_image_source(path, nphotons, seed, counts_out) bind(C, name="ortref_image_source")
        use omp_lib
        character(kind=c_char), intent(IN) :: path(*)
        integer(c_int), value :: nphotons
        integer(c_int64_t), value :: seed
        integer(c_int), intent(OUT) :: counts_out(512, 512)
        integer :: nlocal, nthr
        nthr = omp_get_max_threads()
        call omp_set_num_threads(1)
        call ortref_rng_key(seed, 0, 0_c_int64_t, 0)
        call init_emit_image(cstr(path), imgin_s, nphotons, nlocal)
        call omp_set_num_threads(nthr)
        nphotons_s = nphotons
        source_s = 4
        imgwork_s = imgin_s
        counts_out = imgin_s
    end subroutine ortref_image_source

    subroutine ortref_constants(out) bind(C, name="ortref_constants")
        real(c_double), intent(OUT) :: out(64)
        out = 0.
        out(1)  = bot%nbottle;    out(2)  = bot%ncontents; out(3) = bot%thickness
        out(4)  = bot%radiusa;    out(5)  = bot%radiusb
        out(6)  = bot%centre%x;   out(7)  = bot%centre%y;  out(8) = bot%centre%z
        out(9)  = merge(1., 0., bot%ellipse)
        out(10) = L2a%n1; out(11) = L2a%n2; out(12) = L2b%n2
        out(13) = L2a%centre%z; out(14) = L2a%curve_radius; out(15) = L2a%thickness
        out(16) = L2a%radius; out(17) = L2a%fb; out(18) = L2a%f
        out(19) = L3a%n1; out(20) = L3a%n2; out(21) = L3a%n3; out(22) = L3b%n2; out(23) = L3b%n3
        out(24) = L3a%centre1%z; out(25) = L3a%centre2%z; out(26) = L3a%centre3%z
        out(27) = L3a%R1; out(28) = L3a%R2; out(29) = L3a%R3
        out(30) = L3a%radius; out(31) = L3a%fb; out(32) = L3a%f; out(33) = L3a%thickness
        out(34) = cosThetaMax; out(35) = r1; out(36) = r2; out(37) = img_plane_1
        out(38) = besselDiameter; out(39) = distance
        out(40) = asin(0.22)                               ! imageMod.f90:40
        out(41) = img_diam / 401.                          ! imageMod.f90:45
        out(42) = pi
        out(43) = L3b%centre1%z; out(44) = L3b%centre2%z; out(45) = L3b%centre3%z
        out(46) = L2b%centre%z
        out(47) = spot_size_s
    end subroutine ortref_constants

    ! One ray through the reference path.  status: 0 binned, 1 reached the image
    ! plane but makeImage did not bin it, 3 lost in the bottle, 4 lost in telescope.
    subroutine one_ray(phase, have_in, pos, dir, image, status, xp, yp, emitted, iray)
        integer, intent(IN) :: phase
        logical, intent(IN) :: have_in
        type(vector), intent(INOUT) :: pos, dir
        integer, intent(INOUT) :: image(-200:200, -200:200, 2)
        integer, intent(OUT) :: status, xp, yp
        real, intent(OUT) :: emitted(6)
        integer, intent(IN) :: iray          ! 1-based loop index (create_spot uses it)
        type(stack) :: tracker
        integer(int64) :: cnt
        logical :: skip
        integer :: i, j, i0, j0
        real :: binwid

        skip = .false.
        cnt = 0_int64
        xp = -9999; yp = -9999
        if (phase == 1) then
            if (.not. have_in) then                                          ! main.f90:95-101
                if (source_s == 3) then
                    call iSORS(pos, dir, bot, L2a, isors_offset_s, ring_width_s, .true.)
                elseif (source_s == 5) then        ! test-only: the variant no call site of main.f90 uses
                    call iSORS(pos, dir, bot, L2a, isors_offset_s, ring_width_s, .false.)
                elseif (source_s == 2) then
                    call point_on_bottle(pos, dir, cosThetaMax, bot, spot_size_s)
                else
                    call ring(pos, dir, L2a, r1, r2, bot%radiusa, bot%radiusb, bot%ellipse, bot%centre%z)
                end if
            end if
            emitted = [pos%x, pos%y, pos%z, dir%x, dir%y, dir%z]
            call telescope(pos, dir, L2a, L3a, img_plane_1, cnt, tracker, 0, skip) ! main.f90:104
        else
            if (.not. have_in) then                                          ! main.f90:132-142
                if (source_s == 4) then
                    call emit_image(imgwork_s, pos, dir, L2b)
                elseif (source_s == 1) then
                    call create_spot(pos, dir, cosThetaMax, nphotons_s, iray)
                elseif (source_s == 3) then
                    call point(pos, dir, cosThetaMax, bot%centre%z)
                else
                    call point(pos, dir, cosThetaMax)
                end if
            end if
            emitted = [pos%x, pos%y, pos%z, dir%x, dir%y, dir%z]
            if (use_bottle_s) then
                call bot%forward(pos, dir, tracker, skip)                    ! main.f90:146
            end if
            if (skip) then                                                   ! main.f90:150
                status = 3
                return
            end if
            call telescope(pos, dir, L2b, L3b, img_plane_1, cnt, tracker, 0, skip) ! main.f90:157
        end if
        if (skip) then
            status = 4
            return
        end if
        call makeImage(image, dir, pos, img_diam, phase)                     ! main.f90:108,161
        ! find the bin makeImage incremented (look-up only; the decision is the reference's)
        status = 1
        binwid = img_diam / 401.
        if (abs(pos%x) < 1. .and. abs(pos%y) < 1.) then
            i0 = floor(pos%x / binwid); j0 = floor(pos%y / binwid)
            do j = max(-200, j0-1), min(200, j0+1)
                do i = max(-200, i0-1), min(200, i0+1)
                    if (image(i, j, phase) /= 0) then
                        status = 0; xp = i; yp = j
                        image(i, j, phase) = 0
                    end if
                end do
            end do
        end if
    end subroutine one_ray

    ! Parity entry: explicit rays and/or explicit uniforms, per-ray outputs.
    ! SoA layout [6][n]: x,y,z,dx,dy,dz.  u is [nu][n]; draw k of ray i = u(k*n+i).
    subroutine ortref_trace_rays(phase, n, have_in, pos_dir_in, nu, u, draw_base, &
                                 pos_dir_out, emitted_out, status, bin_xy, ndraws) &
                                 bind(C, name="ortref_trace_rays")
        integer(c_int), value :: phase, have_in, nu, draw_base
        integer(c_int64_t), value :: n
        real(c_double), intent(IN), target :: pos_dir_in(n, 6), u(n, nu)
        real(c_double), intent(OUT) :: pos_dir_out(n, 6), emitted_out(n, 6)
        integer(c_int), intent(OUT) :: status(n), bin_xy(n, 2), ndraws(n)
        integer, allocatable :: image(:, :, :)
        type(vector) :: pos, dir
        integer(c_int64_t) :: i
        integer :: st, xp, yp
        real :: em(6)

        allocate(image(-200:200, -200:200, 2))
        image = 0
        imgwork_s = imgin_s          ! the histogram walk restarts with every call (serial order)
        do i = 1, n
            call ortref_rng_table(c_loc(u(i, 1)), n, nu, draw_base)
            if (have_in /= 0) then
                pos = vector(pos_dir_in(i, 1), pos_dir_in(i, 2), pos_dir_in(i, 3))
                dir = vector(pos_dir_in(i, 4), pos_dir_in(i, 5), pos_dir_in(i, 6))
            end if
            call one_ray(phase, have_in /= 0, pos, dir, image, st, xp, yp, em, int(i))
            pos_dir_out(i, :) = [pos%x, pos%y, pos%z, dir%x, dir%y, dir%z]
            emitted_out(i, :) = em
            status(i) = st
            bin_xy(i, 1) = xp; bin_xy(i, 2) = yp
            ndraws(i) = ortref_rng_draws()
        end do
    end subroutine ortref_trace_rays

    ! Bulk entry: the reference loop bodies (main.f90:90-109, :127-162) over
    ! global ray indices [first, first+n) with ORT-RNG-v2 keyed draws, OpenMP
    ! over rays exactly as main.f90:83-89 (atomic image, reduction on the counter).
    subroutine ortref_trace(phase, first, n, seed, image, lost) bind(C, name="ortref_trace")
        integer(c_int), value :: phase
        integer(c_int64_t), value :: first, n, seed
        integer(c_int), intent(INOUT) :: image(-200:200, -200:200, 2)
        integer(c_int64_t), intent(INOUT) :: lost
        type(vector) :: pos, dir
        type(stack) :: tracker
        integer(int64) :: cnt
        integer(c_int64_t) :: i
        logical :: skip

        cnt = 0_int64
!$omp parallel do default(shared) private(pos, dir, skip, tracker) reduction(+:cnt) schedule(static)
        do i = first, first + n - 1
            skip = .false.
            call ortref_rng_key(seed, phase, i, 0)
            if (phase == 1) then
                if (source_s == 3) then
                    call iSORS(pos, dir, bot, L2a, isors_offset_s, ring_width_s, .true.)
                elseif (source_s == 5) then        ! test-only: the variant no call site of main.f90 uses
                    call iSORS(pos, dir, bot, L2a, isors_offset_s, ring_width_s, .false.)
                elseif (source_s == 2) then
                    call point_on_bottle(pos, dir, cosThetaMax, bot, spot_size_s)
                else
                    call ring(pos, dir, L2a, r1, r2, bot%radiusa, bot%radiusb, bot%ellipse, bot%centre%z)
                end if
                call telescope(pos, dir, L2a, L3a, img_plane_1, cnt, tracker, 0, skip)
                if (skip) cycle
                call makeImage(image, dir, pos, img_diam, 1)
            else
                if (source_s == 1) then
                    call create_spot(pos, dir, cosThetaMax, nphotons_s, int(i + 1))
                elseif (source_s == 3) then
                    call point(pos, dir, cosThetaMax, bot%centre%z)
                else
                    call point(pos, dir, cosThetaMax)
                end if
                if (use_bottle_s) call bot%forward(pos, dir, tracker, skip)
                if (skip) then
                    cnt = cnt + 1_int64
                    cycle
                end if
                call telescope(pos, dir, L2b, L3b, img_plane_1, cnt, tracker, 0, skip)
                if (skip) cycle
                call makeImage(image, dir, pos, img_diam, 2)
            end if
        end do
!$omp end parallel do
        lost = lost + cnt
    end subroutine ortref_trace

end module ortref_api
