! TEST INFRASTRUCTURE — not part of the product.
!
! Substitute for the reference's module `random` (reference src/random_mod.f90:6
! exports ran2, ranu, rang, init_rng) used ONLY when the reference's own path
! sources are compiled into oracle/_ref/libort_ref.so.  The reference's ran2()
! wraps the compiler runtime's random_number, whose stream is compiler- and
! thread-count specific; this module keeps the documented semantics (one U[0,1)
! per call, ranu(a,b) = a + ran2()*(b-a)) but takes the draw from
! oracle/ref/ref_rng.c so that a caller controls, per ray, exactly which
! uniforms the reference code sees.
module random

    use iso_c_binding, only : c_double

    implicit none

    private
    public :: ran2, ranu, rang, init_rng

    interface
        function ortref_draw() bind(C, name="ortref_draw") result(u)
            import :: c_double
            real(c_double) :: u
        end function ortref_draw
    end interface

contains

    subroutine init_rng(input_seed)
        integer, optional, intent(IN) :: input_seed
        ! keyed per ray by the driver; nothing to seed here
    end subroutine init_rng

    real function ran2()
        ran2 = ortref_draw()
    end function ran2

    real function ranu(a, b)
        real, intent(IN) :: a, b
        ranu = a + ran2() * (b - a)
    end function ranu

    subroutine rang(x, y, avg, sigma)
        real, intent(IN)  :: avg, sigma
        real, intent(OUT) :: x, y
        real :: s, const
        s = 1.
        do while (s >= 1.)
            x = ranu(-1., 1.)
            y = ranu(-1., 1.)
            s = y**2 + x**2
        end do
        const = sqrt(-2.*log(s)/s)
        x = avg + sigma*(x*const)
        y = avg + sigma*(y*const)
    end subroutine rang

end module random
